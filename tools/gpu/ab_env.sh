# A/B of an environment switch of the shipped library on ONE box, interleaved: default bench line (no CPU baseline)
# usage: ab_env.sh <tag> <VAR=value> [bench arguments]
export TMPDIR=/tmp
R=$PWD
TAG=${1:-ab}; SW=${2:-CHMC_NO_FIX_IN_JW=1}; shift; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; rm -rf $O/*
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs "$@" > $O/default_$rep.json 2> $O/e.log || tail -3 $O/e.log
  env $SW timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs "$@" > $O/switch_$rep.json 2> $O/e.log || tail -3 $O/e.log
done
O=$O python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.environ['O'] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        kc = c['kernel_classes_warmup']
        print(os.path.basename(f), round(d['value']), round(d['ms_per_step'], 3), (c.get('value_repeats') or {}).get('values'), {k: v['ms_per_step'] for k, v in kc.items() if k in ('jacob_vec', 'elementwise', 'update')})
    except Exception as e:
        print(f, 'unreadable', e)
PY
