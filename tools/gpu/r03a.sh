set -e
mkdir -p gpurun_out/r03a
python tools/sir_iter_hist.py 256 0.25 > gpurun_out/r03a/sir_iter_hist.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r03a/bench_fhn.json 2> gpurun_out/r03a/bench_fhn.err
python bench.py --no-cpu-baseline --config sir > gpurun_out/r03a/bench_sir.json 2> gpurun_out/r03a/bench_sir.err
python tools/iter_hist.py 256 > gpurun_out/r03a/fhn_iter_hist.log 2>&1
echo done
