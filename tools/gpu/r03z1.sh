# round 3 final-build measurements, part 1: GPU test suite, bench lines of every configuration, kernel-trace statistics
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03z
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_fhn_noisy.json 2> $O/bench_fhn_noisy.err || tail -5 $O/bench_fhn_noisy.err
python bench.py --config sir > $O/bench_sir.json 2> $O/bench_sir.err || tail -5 $O/bench_sir.err
python bench.py --config fhn_noiseless --no-cpu-baseline > $O/bench_fhn_noiseless.json 2> $O/bench_fhn_noiseless.err
python bench.py --solver quasi-newton --no-cpu-baseline > $O/bench_fhn_noisy_qn.json 2> $O/bench_qn.err
python bench.py --splitting gaussian --no-cpu-baseline > $O/bench_fhn_noisy_gauss.json 2> $O/bench_gauss.err
python bench.py --config sir --chains-per-gpu 1024 --no-cpu-baseline > $O/bench_sir_1024.json 2> $O/bench_sir_1024.err
python bench.py --num-steps-per-obs 800 --chains-per-gpu 512 --no-cpu-baseline > $O/bench_fhn_noisy_s800_512.json 2> $O/bench_s800.err
python bench.py --engine async --no-cpu-baseline > $O/bench_fhn_noisy_engine_async.json 2> $O/bench_async.err
python bench.py --engine async --config sir --no-cpu-baseline > $O/bench_sir_engine_async.json 2> $O/bench_sir_async.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $O/prof_stats.log 2>&1; cd $R
cp $(find $O/prof_stats -name "*kernel_stats.csv") $O/kernel_stats.csv
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_sir -- python3 $R/bench.py --no-cpu-baseline --config sir > $O/prof_stats_sir.log 2>&1; cd $R
cp $(find $O/prof_stats_sir -name "*kernel_stats.csv") $O/kernel_stats_sir.csv
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_s800 -- python3 $R/bench.py --no-cpu-baseline --num-steps-per-obs 800 --chains-per-gpu 512 > $O/prof_stats_s800.log 2>&1; cd $R
cp $(find $O/prof_stats_s800 -name "*kernel_stats.csv") $O/kernel_stats_s800_512.csv
find $O/prof_stats $O/prof_stats_sir $O/prof_stats_s800 -name "*.csv" -size +4M -delete
python - <<'PY'
import json, glob, pandas as pd
for f in sorted(glob.glob('gpurun_out/r03z/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']; c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), r['kernel'], r['bound'], round(r['frac'], 3),
              'launches', c['launches_per_step'], 'rounds', c['newton_rounds_per_step'], (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'ERR', e)
k = pd.read_csv('gpurun_out/r03z/kernel_stats.csv')
k['name'] = k['Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 70)
print(k[['name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage']].head(14).to_string())
PY
