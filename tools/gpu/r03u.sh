export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03u; mkdir -p $O; rm -rf $O/*
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "config or golden or distinct or status or halves" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1; cd $R
python - <<'PY'
import glob, pandas as pd
k = pd.read_csv(glob.glob('gpurun_out/r03u/prof/**/*kernel_stats.csv', recursive=True)[0])
k['name'] = k['Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 60)
print(k[['name', 'Calls', 'AverageNs', 'Percentage']].head(8).to_string())
PY
rm -rf $O/prof
for r in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$r.json 2> $O/e.log; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03u/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3))
PY
