# round 3, part f: scan LDS pad, device Adam, engine opt-in; full GPU suite + benches
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03f
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
timeout -k 10 300 python tools/adam_timing.py 1024 > $O/adam_1024.log 2>&1; tail -2 $O/adam_1024.log
timeout -k 10 300 python tools/adam_timing.py 256 1 > $O/adam_256.log 2>&1; tail -3 $O/adam_256.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn.json 2> $O/bench_fhn.err || tail -5 $O/bench_fhn.err
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
timeout -k 10 300 python bench.py --no-cpu-baseline --chains-per-gpu 512 > $O/bench_fhn_512_s400.json 2> $O/b1.err || tail -5 $O/b1.err
timeout -k 10 300 python bench.py --no-cpu-baseline --num-steps-per-obs 800 > $O/bench_fhn_256_s800.json 2> $O/b2.err || tail -5 $O/b2.err
timeout -k 10 300 python bench.py --no-cpu-baseline --num-steps-per-obs 800 --chains-per-gpu 512 > $O/bench_fhn_512_s800.json 2> $O/b3.err || tail -5 $O/b3.err
CHMC_SCAN_LDS_PAD=0 timeout -k 10 300 python bench.py --no-cpu-baseline --num-steps-per-obs 800 --chains-per-gpu 512 > $O/bench_fhn_512_s800_nopad.json 2> $O/b4.err || tail -5 $O/b4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03f/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']
        t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches/step', c['launches_per_step'], 'rounds/step', c['newton_rounds_per_step'],
              'constr ms/launch', t['constr']['ms_per_launch'], 'update', t['update']['ms_per_launch'])
    except Exception as e:
        print(f, 'ERR', e)
PY
