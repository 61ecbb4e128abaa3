# A/B of the scan hand-over depth (NB = 1 default build against build/libchmc_NB2.so), interleaved runs
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03w; mkdir -p $O
for rep in 1 2 3; do
  for m in NB1 NB2; do
    lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so; [ $m = NB2 ] && lib=$R/build/libchmc_NB2.so
    CHMC_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_${m}_r$rep.json 2> $O/err.log || tail -3 $O/err.log
  done
done
for m in NB1 NB2; do
  lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so; [ $m = NB2 ] && lib=$R/build/libchmc_NB2.so
  CHMC_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --num-steps-per-obs 800 --chains-per-gpu 512 > $O/bench_${m}_s800.json 2> $O/err.log || tail -3 $O/err.log
  CHMC_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --config fhn_noiseless > $O/bench_${m}_noiseless.json 2> $O/err.log || tail -3 $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03w/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; t = c['kernel_classes_warmup']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'constr us/launch', round(t['constr']['ms_per_launch']*1e3,1))
PY
