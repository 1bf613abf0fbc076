# Usage (through gpurun): bash tools/micro/prof_one.sh <tag> <python script> [args...] : kernel trace + stats into gpurun_out/<tag>/
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o $TAG -- python "$@" > gpurun_out/$TAG/out.txt 2> gpurun_out/$TAG/err.txt
ls gpurun_out/$TAG | head
