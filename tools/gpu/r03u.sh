export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or row_split" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export CHMC_GLD16_ROWS=1; else unset CHMC_GLD16_ROWS; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_rows$v.json 2> $O/e.log || tail -3 $O/e.log
done
unset CHMC_GLD16_ROWS
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024.json 2> $O/e.log || tail -3 $O/e.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u/bench_sir_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; t = c['kernel_classes_warmup']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'launches', c['launches_per_step'], 'gld', t.get('grad_log_det_blk', {}).get('ms_per_step'))
PY
