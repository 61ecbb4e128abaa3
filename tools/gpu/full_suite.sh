# the whole GPU suite as the driver runs it (one process), log under gpurun_out/<tag>/
export TMPDIR=/tmp
TAG=${1:-suite}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
