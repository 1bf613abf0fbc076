cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcq -o q1 -- python tools/passq_bench.py 500000 10000 "512 5 6 4 1" > gpurun_out/pmcq/out1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmcq -o q2 -- python tools/passq_bench.py 500000 10000 "512 5 6 4 1" > gpurun_out/pmcq/out2.txt 2>&1
ls gpurun_out/pmcq
