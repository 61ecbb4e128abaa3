export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03m
mkdir -p $O
for m in NB2 NB4 PRIO DEPTH5 DEPTH3; do
  CHMC_HIP_LIBRARY=$R/build/libchmc_$m.so timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$m.json 2> $O/b_$m.err || tail -5 $O/b_$m.err
done
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/b_default.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03m/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['config']; t = c['kernel_classes_warmup']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'constr us/launch', round(t['constr']['ms_per_launch']*1e3,1))
    except Exception as e:
        print(f, 'ERR', e)
PY
