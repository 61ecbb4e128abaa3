export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O; rm -f $O/*
CHMC_STATE_TWO_PHASE=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "config or ops or golden or distinct or status" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_lean_$r.json 2> $O/e.log
  CHMC_STATE_TWO_PHASE=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_two_$r.json 2> $O/e.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); t = d['config']['kernel_classes_warmup']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'state_blk', t.get('state_blk'))
PY
