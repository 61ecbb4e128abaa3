export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp
for wv in 2; do
  CHMC_PAR_WAVES=$wv timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_w$wv -- python3 $R/bench.py --no-cpu-baseline --config sir > $O/tr_w$wv.log 2>&1 || tail -5 $O/tr_w$wv.log
done
cd $R
python - <<'PY'
import glob, pandas as pd, numpy as np
for wv in (2,):
    f = glob.glob(f'gpurun_out/r03p/tr_w{wv}/**/*kernel_trace.csv', recursive=True)[0]
    d = pd.read_csv(f).sort_values('Start_Timestamp')
    d['dur'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
    last_adam = d[d.Kernel_Name.str.contains('k_nld_grad_wave')].End_Timestamp.max()
    s = d[d.Start_Timestamp > last_adam]
    # the timed region: the last 32 steps; take the last 60 % of the launches after init
    s = s.iloc[int(len(s) * 0.4):]
    p = s[s.Kernel_Name.str.contains('k_fwd_par')]
    print(f'W={wv}: {len(s)} launches after init (tail), span {(s.End_Timestamp.max()-s.Start_Timestamp.min())/1e6:.1f} ms, sum {s.dur.sum()/1e3:.1f} ms; k_fwd_par {len(p)} launches, sum {p.dur.sum()/1e3:.1f} ms, mean {p.dur.mean():.1f} us')
    print('  fwd_par duration quantiles us', np.round(np.quantile(p.dur, [0.05, 0.25, 0.5, 0.75, 0.9, 0.99]), 1).tolist())
    print('  hist', np.histogram(p.dur, bins=[0, 25, 50, 75, 100, 150, 200, 300, 400, 600, 1000, 5000])[0].tolist())
    g = s.groupby(s.Kernel_Name.str.slice(0, 60)).dur.agg(['size', 'sum', 'mean']).sort_values('sum', ascending=False)
    print(g.head(12).to_string())
    d.to_csv(f'gpurun_out/r03p/trace_w{wv}.csv.gz', index=False, columns=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp']) if False else None
PY
rm -rf $O/tr_w2
