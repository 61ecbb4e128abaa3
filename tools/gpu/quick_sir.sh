# targeted SIR tests + three batch sizes with the default switches + phase profile (diagnostic build, if present) + Adam finder
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r04n}; mkdir -p $O; rm -rf $O/*
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sir or Sir or shard or adam or per_chain or trajector or parallel" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for b in 256 512 1024; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --config sir --chains-per-gpu $b > $O/bench_sir_${b}.json 2> $O/e1.log || tail -5 $O/e1.log
done
O=$O python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.environ['O'] + '/bench_sir*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches', c.get('launches_per_step'), (c.get('value_repeats') or {}).get('values'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
if [ -f $R/build/libchmc_prof.so ]; then CHMC_HIP_LIBRARY=$R/build/libchmc_prof.so timeout -k 10 300 python tools/retract_prof.py 256 2 > $O/prof256.log 2>&1; tail -34 $O/prof256.log; fi
for s in 1.0 variable; do timeout -k 10 300 python tools/adam_timing.py 1024 0 $s > $O/adam_$s.log 2>&1; tail -2 $O/adam_$s.log; done
