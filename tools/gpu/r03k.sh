export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03k
mkdir -p $O
for m in 10 12 16; do
  CHMC_HIP_LIBRARY=$R/build/libchmc_maxs$m.so timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_maxs$m.json 2> $O/b$m.err || tail -5 $O/b$m.err
done
timeout -k 10 300 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_maxs6.json 2> $O/b6.err || tail -5 $O/b6.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03k/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']; t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds/step', c['newton_rounds_per_step'], 'constr us/launch', t['constr']['ms_per_launch']*1e3)
    except Exception as e:
        print(f, 'ERR', e)
PY
