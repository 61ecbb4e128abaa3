# round 3, part d: engine with merged Newton launches -- parity, benches, kernel trace
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_async_engine.py -m gpu -x -q > $O/pytest_async.log 2>&1; echo "pytest rc $?" >> $O/pytest_async.log
tail -5 $O/pytest_async.log
grep -q "pytest rc 0" $O/pytest_async.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn_async.json 2> $O/bench_fhn_async.err || tail -5 $O/bench_fhn_async.err
timeout -k 10 300 python bench.py --no-cpu-baseline --lockstep > $O/bench_fhn_lock.json 2> $O/bench_fhn_lock.err || tail -5 $O/bench_fhn_lock.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_async.json 2> $O/bench_sir_async.err || tail -5 $O/bench_sir_async.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --lockstep > $O/bench_sir_lock.json 2> $O/bench_sir_lock.err || tail -5 $O/bench_sir_lock.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_fhn -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 32 --warmup 16 > $O/trace_fhn.log 2>&1)
python tools/trace_overlap.py $O/trace_fhn 0.25 > $O/trace_fhn_overlap.txt 2>&1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_sir -- python3 $R/bench.py --no-cpu-baseline --no-profile --config sir --steps 32 --warmup 16 > $O/trace_sir.log 2>&1)
python tools/trace_overlap.py $O/trace_sir 0.25 > $O/trace_sir_overlap.txt 2>&1
find $O -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03d/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches/step', c['launches_per_step'], 'rounds/step', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'], 3), 'attempted', c['chain_steps_attempted'])
    except Exception as e:
        print(f, 'ERR', e)
PY
