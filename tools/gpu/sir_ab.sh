export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04i; mkdir -p $O; rm -rf $O/*
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or Sir or parallel or shard or adam or row_split or other_baseline or trajector or per_chain" > $O/pytest_sir.log 2>&1 || { tail -60 $O/pytest_sir.log; exit 1; }
tail -2 $O/pytest_sir.log
for b in 256 1024; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --config sir --chains-per-gpu $b > $O/bench_sir_$b.json 2> $O/e1.log || tail -5 $O/e1.log
  CHMC_RETRACT_KERNEL=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --config sir --chains-per-gpu $b > $O/bench_sir_${b}_batched.json 2> $O/e1.log || tail -5 $O/e1.log
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r04i/bench_sir*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c.get('newton_rounds_per_step'), 'ok', round(c['step_success_rate'], 4), 'launches', c.get('launches_per_step'), (c.get('value_repeats') or {}).get('values'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
if [ -f $R/build/libchmc_prof.so ]; then CHMC_HIP_LIBRARY=$R/build/libchmc_prof.so timeout -k 10 300 python tools/retract_prof.py 256 2 > $O/prof256.log 2>&1; tail -32 $O/prof256.log; fi
