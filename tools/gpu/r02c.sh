set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02c
for H in 1 2; do
  cd /tmp && CHMC_HALVES=$H rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02c/trace_h$H -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 6 --warmup 2 --burn-iters 2 > $R/gpurun_out/r02c/trace_h$H.log 2>&1
  cd $R
  python tools/trace_overlap.py gpurun_out/r02c/trace_h$H 0.25 > gpurun_out/r02c/overlap_h$H.txt 2>&1
  find gpurun_out/r02c/trace_h$H -name "*.csv" -size +20M -delete
done
cat gpurun_out/r02c/overlap_h1.txt gpurun_out/r02c/overlap_h2.txt
