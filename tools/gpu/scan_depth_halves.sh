# forward-scan ring depth (tiles in flight) x half-batch overlap at configs[1] and S = 800 x 512: the shipped library (depth 4)
# against build/libchmc_d6.so / libchmc_d8.so (-DCHMC_SCAN_DEPTH=(V==2?6:3) / 8), with and without CHMC_HALVES=2
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r04u}; mkdir -p $O; rm -rf $O/*
for lib in default d6 d8; do
  L=""; if [ $lib != default ]; then L=$R/build/libchmc_$lib.so; [ -f $L ] || continue; fi
  for h in 1 2; do
    for shape in "" "--num-steps-per-obs 800 --chains-per-gpu 512"; do
      tag=${lib}_h${h}_$( [ -z "$shape" ] && echo s400 || echo s800 )
      env ${L:+CHMC_HIP_LIBRARY=$L} $( [ $h = 2 ] && echo CHMC_HALVES=2 ) timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs $shape > $O/$tag.json 2> $O/e.log || tail -3 $O/e.log
    done
  done
done
O=$O python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.environ['O'] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        kc = c['kernel_classes_warmup']
        print(os.path.basename(f), round(d['value']), round(d['ms_per_step'], 3), (c.get('value_repeats') or {}).get('values'), 'constr', kc['constr']['ms_per_launch'], 'update', kc['update']['ms_per_launch'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
