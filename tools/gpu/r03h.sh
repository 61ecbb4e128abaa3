# round 3, part h: pruned kernel families, device-resident doubling; full GPU suite, dynamic timing, benches
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03h
mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python tools/dynamic_timing.py > $O/dynamic_timing.log 2>&1; cat $O/dynamic_timing.log; timeout -k 10 300 python tools/dynamic_timing.py 256 400 24 > $O/dynamic_timing_24.log 2>&1; head -2 $O/dynamic_timing_24.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn.json 2> $O/bench_fhn.err || tail -5 $O/bench_fhn.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03h/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches/step', c['launches_per_step'], 'rounds/step', c['newton_rounds_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
