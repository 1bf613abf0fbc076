#!/bin/bash
# Round profile on the GPU box.  part 1: the default bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM
# traffic, MFMA busy, waits, LDS conflicts) on three launches of each hot kernel at the headline shard shape.  part 2: kernel
# stats + PMC passes for the rest of the path (feature forward C4 / FPN, RLS chain, one Minibootstrap).
# Usage (through gpurun): bash tools/profile_round.sh rNN [1|2]
set -u
R=${1:-r01}
PART=${2:-1}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMC1="FETCH_SIZE"
PMC2="WRITE_SIZE"
PMC3="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PMC4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"
if [ "$PART" = "1" ]; then
  # the kernel sources these passes are taken on: bench.py quotes `roofline.traffic` only from passes of ITS tree's sources
  python -c "import json, bench; print(json.dumps({'csrc_sha16': bench.csrc_sha16()}))" > $OUT/pmc_meta.json
  python bench.py --steps 3 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
  for c in "$PMC1" "$PMC2" "$PMC3" "$PMC4"; do
    tag=pmc_$(echo $c | cut -d' ' -f1)
    ODX_N=1000000 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o $tag -- python tools/prof_kernels.py > /dev/null 2>&1
  done
  rm -f $OUT/bench_kernel_trace.csv
else
  for job in "rls tools/prof_rls.py" "forward tools/prof_forward.py both 10" "forward_b4 tools/prof_forward_batch.py 4 f32" "forward_b4_bf16 tools/prof_forward_batch.py 4 bf16" "forward_b8 tools/prof_forward_batch.py 8 f32" "forward_fpn_b8 tools/prof_forward_batch.py 8 f32 fpn" "minibootstrap tools/prof_minibootstrap.py"; do
    set -- $job
    tag=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o x_$tag -- python "$@" > $OUT/x_$tag.out 2> $OUT/x_$tag.err
  done
  for job in "rls tools/prof_rls.py" "forward tools/prof_forward.py both 4"; do
    set -- $job
    tag=$1; shift
    for c in "$PMC1" "$PMC2" "$PMC3"; do
      ptag=xpmc_${tag}_$(echo $c | cut -d' ' -f1)
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o $ptag -- python "$@" > /dev/null 2>&1
    done
  done
fi
rm -f $OUT/*agent_info.csv $OUT/*domain_stats.csv
python tools/summarize_profile.py $OUT > $OUT/summary.md
tail -n 40 $OUT/summary.md
