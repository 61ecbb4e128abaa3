# 4-wavefront per-chain kernels (two chains per CU) against the 8-wavefront ones and the batched path
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r04m}; mkdir -p $O; rm -rf $O/*
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "per_chain_kernels or adam" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for b in 256 512 1024 2048; do
  for k in 1 2 0; do
    CHMC_RETRACT_KERNEL=$k timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --config sir --chains-per-gpu $b > $O/bench_sir_${b}_k$k.json 2> $O/e1.log || tail -5 $O/e1.log
  done
done
O=$O python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.environ.get('O', 'gpurun_out/r04m') + '/bench_sir*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'launches', c.get('launches_per_step'), (c.get('value_repeats') or {}).get('values'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for s in 1.0 variable; do timeout -k 10 300 python tools/adam_timing.py 1024 0 $s > $O/adam_$s.log 2>&1; tail -1 $O/adam_$s.log; done
