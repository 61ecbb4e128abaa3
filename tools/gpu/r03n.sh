# multi-wavefront time-parallel scan: parity first, then the SIR bench per W
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03n
mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "sir or parallel_scan or Sir or adam" > $O/pytest_sir.log 2>&1 || { tail -30 $O/pytest_sir.log; exit 1; }
tail -3 $O/pytest_sir.log
for wv in 1 2 4; do
  CHMC_PAR_WAVES=$wv timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_w$wv.json 2> $O/b_sir_w$wv.err || tail -5 $O/b_sir_w$wv.err
done
CHMC_PAR_WAVES=2 timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 512 > $O/bench_sir512_w2.json 2> $O/b2.err
CHMC_PAR_WAVES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 512 > $O/bench_sir512_w1.json 2> $O/b3.err
timeout -k 10 300 python tools/adam_timing.py 256 1 > $O/adam256.log 2>&1; tail -3 $O/adam256.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03n/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']; t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'constr us/launch', round(t['constr']['ms_per_launch']*1e3,1), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'],4))
    except Exception as e:
        print(f, 'ERR', e)
PY
