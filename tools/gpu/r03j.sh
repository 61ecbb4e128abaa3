export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03j
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_async_engine.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --engine async > $O/bench_sir_async.json 2> $O/bench_sir_async.err || tail -5 $O/bench_sir_async.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024.json 2> $O/b.err || tail -5 $O/b.err
timeout -k 10 300 python tools/par_scan_stats.py 256 > $O/par_scan_stats.log 2>&1; tail -5 $O/par_scan_stats.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03j/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']; t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches/step', c['launches_per_step'], 'rounds/step', c['newton_rounds_per_step'], 'constr', t['constr'])
    except Exception as e:
        print(f, 'ERR', e)
PY
