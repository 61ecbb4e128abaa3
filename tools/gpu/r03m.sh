# run-to-run spread of the default bench line on one box
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O; rm -f $O/*
for r in 1 2 3 4; do timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$r.json 2> $O/e.log || tail -3 $O/e.log; done
timeout -k 10 300 python bench.py > $O/bench_full.json 2> $O/e.log || tail -3 $O/e.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03m/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'frac', round(r['frac'], 3), 'scan ms', round(r['avg_launch_ms'], 4), 'traffic', r.get('traffic') is not None)
PY
