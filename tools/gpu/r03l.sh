export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03l
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn.json 2> $O/bench_fhn.err || tail -5 $O/bench_fhn.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03l/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']; t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds/step', c['newton_rounds_per_step'], 'launches', c['launches_per_step'])
        for k, v in sorted(t.items(), key=lambda kv: -kv[1]['ms_per_step']):
            print('   %-18s %6.3f ms/step %6.1f launches %7.1f us' % (k, v['ms_per_step'], v['launches_per_step'], v['ms_per_launch'] * 1e3))
    except Exception as e:
        print(f, 'ERR', e)
PY
