export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O; rm -rf $O/*
for wv in 1 2 4; do
  CHMC_PAR_WAVES=$wv timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_w$wv.json 2> $O/e.log
  CHMC_PAR_WAVES=$wv timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 512 > $O/bench512_w$wv.json 2> $O/e.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u/bench*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'])
PY
