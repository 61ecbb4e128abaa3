export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03t; mkdir -p $O; rm -f $O/*
for m in base WILD30 WILD5; do
  lib=$R/build/libchmc_$m.so; [ $m = base ] && lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so
  CHMC_HIP_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_$m.json 2> $O/err_$m.log || tail -3 $O/err_$m.log
  CHMC_HIP_LIBRARY=$lib timeout -k 10 200 python tools/par_scan_stats.py 256 200 > $O/stats_$m.log 2>&1
done
python - <<'PY'
import json, glob, re
for f in sorted(glob.glob('gpurun_out/r03t/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'],4))
for f in sorted(glob.glob('gpurun_out/r03t/stats_*.log')):
    print(f.split('/')[-1], [re.search(r'own-previous-iterate \[(.*?)\]', l).group(1).split(', ')[-1] for l in open(f) if 'traj' in l])
PY
