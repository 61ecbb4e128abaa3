export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O; rm -rf $O/*
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or Sir or time_parallel or adam or switch_partition or unconstrained" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_$r.json 2> $O/e.log; done
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir1024.json 2> $O/e.log
timeout -k 10 300 python tools/adam_timing.py 1024 > $O/adam.log 2>&1; tail -1 $O/adam.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; t = c['kernel_classes_warmup']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'constr us', round(t['constr']['ms_per_launch'] * 1e3, 1))
PY
