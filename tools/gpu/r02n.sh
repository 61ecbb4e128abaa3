export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02n
CHMC_PAR_SCAN=1 python tools/par_scan_stats.py 256 200 0.25 | cut -c1-460
python tools/par_scan_compare.py 256 200 0.25 12 | cut -c1-500
for P in 0 1; do
CHMC_PAR_SCAN=$P python bench.py --config sir --no-cpu-baseline > gpurun_out/r02n/bench_sir_par$P.json 2>/dev/null
python - $P <<'PY'
import json,sys
P=sys.argv[1]
d=json.loads(open(f'gpurun_out/r02n/bench_sir_par{P}.json').read().strip().splitlines()[-1])
print('PAR_SCAN',P,'steps/s',round(d['value']),'ms',round(d['ms_per_step'],2),'succ',d['config']['step_success_rate'],'k',round(d['config']['mean_newton_iters_fwd_plus_bwd'],3), 'constr', d['config']['kernel_classes_warmup']['constr'])
PY
done
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir or config or full_size" 2>&1 | tail -3
cd /tmp && CHMC_PAR_SCAN=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02n/p -- python3 $R/tools/par_scan_stats.py 256 200 0.05 > $R/gpurun_out/r02n/stats05.log 2>&1; cd $R
python - <<'PY'
import pandas as pd, glob
f=glob.glob('gpurun_out/r02n/p/**/*kernel_trace.csv',recursive=True)[0]
k=pd.read_csv(f).sort_values('Start_Timestamp')
k['dur']=(k['End_Timestamp']-k['Start_Timestamp'])/1e3
k=k.iloc[int(len(k)*0.7):]
d=k[k.Kernel_Name.str.contains('k_fwd_par',regex=False)]['dur']
print(f"k_fwd_par (dt 0.05, no diverging chains) n {len(d)} mean {d.mean():.1f} median {d.median():.1f} min {d.min():.1f} max {d.max():.1f}")
PY
grep "traj 3" gpurun_out/r02n/stats05.log | cut -c1-400
find gpurun_out/r02n -name "*.csv" -size +2M -delete
