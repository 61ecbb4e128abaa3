# quick check of a change on both headline configurations: targeted parity tests, then FHN and SIR bench lines (two runs each)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not notebook and not example" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for r in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_fhn_$r.json 2> $O/e.log || tail -3 $O/e.log
done
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_1.json 2> $O/e.log || tail -3 $O/e.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config fhn_noiseless > $O/bench_noiseless_1.json 2> $O/e.log || tail -3 $O/e.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03y/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; t = c['kernel_classes_warmup']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches', c['launches_per_step'], 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'],4), 'gld ms/step', t.get('grad_log_det_blk', {}).get('ms_per_step'))
PY
