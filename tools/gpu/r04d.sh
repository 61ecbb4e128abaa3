# FHN headline: parity subset, bench (two runs), kernel stats of the bench command
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r04d}; mkdir -p $O; rm -rf $O/*
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_trajectories.py -m gpu -x -q -k "not sir and not Sir" > $O/pytest_fhn.log 2>&1 || { tail -40 $O/pytest_fhn.log; exit 1; }
tail -2 $O/pytest_fhn.log
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_fhn_$i.json 2> $O/e$i.log || tail -5 $O/e$i.log; done
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1; cd $R
cp $(find $O/prof -name "*kernel_stats.csv") $O/kernel_stats.csv 2>/dev/null
find $O/prof -name "*.csv" -size +2M -delete
python - $O <<'PY'
import glob, json, sys, pandas as pd
O = sys.argv[1]
for f in sorted(glob.glob(O + '/bench_fhn_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']; r = d['roofline']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c.get('newton_rounds_per_step'), 'launches', c.get('launches_per_step'), 'roofline', r['bound'], round(r['frac'], 3))
    kc = c.get('kernel_classes_warmup', {})
    print('   ', {k: round(v['ms_per_step'], 3) for k, v in kc.items() if isinstance(v, dict) and 'ms_per_step' in v})
s = pd.read_csv(O + '/kernel_stats.csv')
s['Name'] = s['Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 60)
print(s[['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage']].head(14).to_string())
PY
