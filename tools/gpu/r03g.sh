# round 3, part g: notebook posterior with the dynamic transition (per-chain outcome table) and static; Adam timing
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03g
mkdir -p $O
timeout -k 10 300 python tools/adam_timing.py 1024 > $O/adam_1024.log 2>&1; tail -1 $O/adam_1024.log
timeout -k 10 600 python examples/fhn_notebook_posterior.py 64 450 200 24 $O/nb_dyn dynamic > $O/notebook_dynamic_64x450.log 2>&1; tail -32 $O/notebook_dynamic_64x450.log
rm -rf $O/nb_dyn
