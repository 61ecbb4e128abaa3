set -x
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
python bench.py > gpurun_out/r02a/bench_fhn_noisy.json 2> gpurun_out/r02a/bench_fhn_noisy.err
python bench.py --config fhn_noiseless --no-cpu-baseline > gpurun_out/r02a/bench_fhn_noiseless.json 2> gpurun_out/r02a/bench_fhn_noiseless.err
python bench.py --solver quasi-newton --no-cpu-baseline > gpurun_out/r02a/bench_fhn_noisy_qn.json 2> gpurun_out/r02a/bench_qn.err
python bench.py --splitting gaussian --no-cpu-baseline > gpurun_out/r02a/bench_fhn_noisy_gauss.json 2> gpurun_out/r02a/bench_gauss.err
CHMC_BENCH_VERBOSE=1 python bench.py --config sir --no-cpu-baseline > gpurun_out/r02a/bench_sir.json 2> gpurun_out/r02a/bench_sir.err
tail -c 600 gpurun_out/r02a/pytest.log
