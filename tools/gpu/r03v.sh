export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc = 0 ] || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/e1.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024.json 2> $O/e2.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 128 > $O/bench_sir_128.json 2> $O/e3.log
timeout -k 10 200 python tools/adam_timing.py 1024 > $O/adam1024.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03v/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'],4))
PY
tail -2 $O/adam1024.log
