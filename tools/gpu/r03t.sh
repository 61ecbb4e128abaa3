export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03t; mkdir -p $O
for m in 8 12 16 24; do
  lib=$R/build/libchmc_MAXS$m.so; [ $m = 12 ] && lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so
  for wv in 1 2 4; do
    CHMC_HIP_LIBRARY=$lib CHMC_PAR_WAVES=$wv timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_m${m}_w$wv.json 2> $O/err_m${m}_w$wv.log || tail -3 $O/err_m${m}_w$wv.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03t/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'],4))
    except Exception as e:
        print(f, 'ERR', e)
PY
