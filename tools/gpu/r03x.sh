# SIR (16-row block) state evaluation work: parity tests, bench, kernel-trace summary of the timed part
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03x; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sir or Sir or parallel or adam or mfma" > $O/pytest_sir.log 2>&1 || { tail -30 $O/pytest_sir.log; exit 1; }
tail -2 $O/pytest_sir.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/e1.log || tail -5 $O/e1.log
for ns in 1 2; do
  CHMC_ROW_SPLIT=$ns timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_ns$ns.json 2> $O/e1.log || tail -5 $O/e1.log
done
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024.json 2> $O/e1.log || tail -5 $O/e1.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --config sir > $O/tr.log 2>&1 || tail -5 $O/tr.log
cd $R
python - <<'PY'
import glob, json, pandas as pd, numpy as np
d = json.loads(open('gpurun_out/r03x/bench_sir.json').read().strip().splitlines()[-1]); c = d['config']
print('bench sir', round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'], 4))
for f in sorted(glob.glob('gpurun_out/r03x/bench_sir_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
    print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c['newton_rounds_per_step'], 'ok', round(c['step_success_rate'], 4))
f = glob.glob('gpurun_out/r03x/tr/**/*kernel_trace.csv', recursive=True)[0]
d = pd.read_csv(f).sort_values('Start_Timestamp')
d['dur'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
last_adam = d[d.Kernel_Name.str.contains('k_nld_grad_wave')].End_Timestamp.max()
s = d[d.Start_Timestamp > last_adam]
s = s.iloc[int(len(s) * 0.4):]
print(f'{len(s)} launches (tail), span {(s.End_Timestamp.max()-s.Start_Timestamp.min())/1e6:.1f} ms, sum {s.dur.sum()/1e3:.1f} ms')
g = s.groupby(s.Kernel_Name.str.slice(0, 70)).dur.agg(['size', 'sum', 'mean']).sort_values('sum', ascending=False)
print(g.head(14).to_string())
PY
rm -rf $O/tr
