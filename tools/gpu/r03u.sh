export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O; rm -rf $O/*
for r in 1 2 3; do
  for v in pub nopub; do
    if [ $v = nopub ]; then export CHMC_NO_PUBLISH=1; else unset CHMC_NO_PUBLISH; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_fhn_${v}_$r.json 2> $O/e.log
    timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_${v}_$r.json 2> $O/e.log
  done
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    acc[f.split('/')[-1].rsplit('_', 1)[0]].append(round(d['value']))
for k, v in acc.items(): print(k, v)
PY
