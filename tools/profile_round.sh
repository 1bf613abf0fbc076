#!/bin/bash
# Round profile on the GPU box: (1) the default bench line, (2) rocprofv3 kernel stats of the same
# command, (3) PMC passes (HBM traffic, MFMA busy) on one launch of each hot kernel.
# Usage (through gpurun): bash tools/profile_round.sh rNN
set -u
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=pmc_$(echo $c | cut -d' ' -f1)
  ODX_N=1000000 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o $tag -- python tools/prof_kernels.py > /dev/null 2>&1
done
python tools/summarize_profile.py $OUT > $OUT/summary.md
cat $OUT/summary.md
rm -f $OUT/*agent_info.csv $OUT/bench_kernel_trace.csv
