# round 4, first GPU call: the per-chain retraction kernel (SIR single block) -- parity, A/B against the lock-step rounds, trace
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04a; mkdir -p $O; rm -rf $O/*
# smallest first: a hang must cost seconds, not the box
timeout -k 10 180 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "test_baseline_config4_size_sir_s200" > $O/pytest_first.log 2>&1 || { tail -30 $O/pytest_first.log; exit 1; }
tail -2 $O/pytest_first.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --ignore tests/test_async_engine.py -k "sir or Sir or parallel or shard or half or adam or mfma or row_split or other_baseline" > $O/pytest_sir.log 2>&1 || { tail -40 $O/pytest_sir.log; exit 1; }
tail -2 $O/pytest_sir.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir.json 2> $O/e1.log || tail -5 $O/e1.log
CHMC_RETRACT_KERNEL=0 timeout -k 10 200 python bench.py --no-cpu-baseline --config sir > $O/bench_sir_lockstep.json 2> $O/e1.log || tail -5 $O/e1.log
timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024.json 2> $O/e1.log || tail -5 $O/e1.log
CHMC_RETRACT_KERNEL=0 timeout -k 10 200 python bench.py --no-cpu-baseline --config sir --chains-per-gpu 1024 > $O/bench_sir_1024_lockstep.json 2> $O/e1.log || tail -5 $O/e1.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-cpu-baseline --config sir > $O/tr.log 2>&1 || tail -5 $O/tr.log
cd $R
python - <<'PY'
import glob, json, pandas as pd, numpy as np
for f in sorted(glob.glob('gpurun_out/r04a/bench_sir*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'], 3), 'rounds', c.get('newton_rounds_per_step'), 'ok', round(c['step_success_rate'], 4), 'launches', c.get('launches_per_step'))
    except Exception as e:
        print(f, 'unreadable', e)
f = glob.glob('gpurun_out/r04a/tr/**/*kernel_trace.csv', recursive=True)[0]
d = pd.read_csv(f).sort_values('Start_Timestamp')
d['dur'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
last_adam = d[d.Kernel_Name.str.contains('k_nld_grad_wave')].End_Timestamp.max()
s = d[d.Start_Timestamp > last_adam]
s = s.iloc[int(len(s) * 0.4):]
print(f'{len(s)} launches (tail), span {(s.End_Timestamp.max()-s.Start_Timestamp.min())/1e6:.1f} ms, sum {s.dur.sum()/1e3:.1f} ms')
g = s.groupby(s.Kernel_Name.str.slice(0, 70)).dur.agg(['size', 'sum', 'mean', 'max']).sort_values('sum', ascending=False)
print(g.head(16).to_string())
r = s[s.Kernel_Name.str.contains('k_retract_chain')].dur
print('retract kernel durations us: quantiles', np.quantile(r, [0, .1, .25, .5, .75, .9, 1]).round(1))
PY
rm -rf $O/tr
