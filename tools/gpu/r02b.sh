set -x
mkdir -p gpurun_out/r02b
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "half or full_size or steps_small" > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02b/pytest.log
tail -c 400 gpurun_out/r02b/pytest.log
for i in 1 2; do
CHMC_HALVES=1 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_h1_$i.json 2> gpurun_out/r02b/bench_h1_$i.err
CHMC_HALVES=2 python bench.py --no-cpu-baseline > gpurun_out/r02b/bench_h2_$i.json 2> gpurun_out/r02b/bench_h2_$i.err
done
CHMC_HALVES=2 python bench.py --no-cpu-baseline --no-profile > gpurun_out/r02b/bench_h2_noprof.json 2> gpurun_out/r02b/bench_h2_noprof.err
CHMC_HALVES=1 python bench.py --no-cpu-baseline --no-profile > gpurun_out/r02b/bench_h1_noprof.json 2> gpurun_out/r02b/bench_h1_noprof.err
CHMC_HALVES=2 python bench.py --no-cpu-baseline --config sir > gpurun_out/r02b/bench_sir_h2.json 2> gpurun_out/r02b/bench_sir_h2.err
CHMC_HALVES=2 python bench.py --no-cpu-baseline --chains-per-gpu 512 > gpurun_out/r02b/bench_h2_512.json 2> gpurun_out/r02b/bench_h2_512.err
CHMC_HALVES=1 python bench.py --no-cpu-baseline --chains-per-gpu 512 > gpurun_out/r02b/bench_h1_512.json 2> gpurun_out/r02b/bench_h1_512.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), d['config']['step_success_rate'])
    except Exception as e: print(f,'ERR',e)
PY
