# round 3, part e: fused factor/solve/muF kernel, engine gating; scan time vs (chains, S)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_async_engine.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn_async.json 2> $O/bench_fhn_async.err || tail -5 $O/bench_fhn_async.err
timeout -k 10 300 python bench.py --no-cpu-baseline --lockstep > $O/bench_fhn_lock.json 2> $O/bench_fhn_lock.err || tail -5 $O/bench_fhn_lock.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_async.json 2> $O/bench_sir_async.err || tail -5 $O/bench_sir_async.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --lockstep > $O/bench_sir_lock.json 2> $O/bench_sir_lock.err || tail -5 $O/bench_sir_lock.err
timeout -k 10 300 python bench.py --no-cpu-baseline --lockstep --chains-per-gpu 512 > $O/bench_fhn_lock_512_s400.json 2> $O/b1.err || tail -5 $O/b1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --lockstep --num-steps-per-obs 800 > $O/bench_fhn_lock_256_s800.json 2> $O/b2.err || tail -5 $O/b2.err
timeout -k 10 300 python bench.py --no-cpu-baseline --lockstep --num-steps-per-obs 800 --chains-per-gpu 512 > $O/bench_fhn_lock_512_s800.json 2> $O/b3.err || tail -5 $O/b3.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03e/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']
        t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches/step', c['launches_per_step'], 'rounds/step', c['newton_rounds_per_step'],
              'constr ms/launch', t['constr']['ms_per_launch'], 'update', t['update']['ms_per_launch'], 'solve', t.get('solve_chain', {}).get('ms_per_step'), 'sym', t.get('sym_blk', {}).get('ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
