export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02k
cd /tmp && CHMC_PAR_SCAN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02k/p -- python3 $R/tools/par_scan_stats.py 256 200 0.05 > $R/gpurun_out/r02k/stats.log 2>&1; cd $R
tail -4 gpurun_out/r02k/stats.log
python - <<'PY'
import pandas as pd, glob
f=glob.glob('gpurun_out/r02k/p/**/*kernel_trace.csv',recursive=True)[0]
k=pd.read_csv(f).sort_values('Start_Timestamp')
k['dur']=(k['End_Timestamp']-k['Start_Timestamp'])/1e3
k=k.iloc[int(len(k)*0.7):]
for pat in ('k_fwd_par','k_fwd_scan','k_rev_wave','k_gram_rows','KNewtonFactor','k_gld_bwd','k_gld_fwd'):
    d=k[k.Kernel_Name.str.contains(pat,regex=False)]['dur']
    if len(d): print(f"  {pat:20s} n {len(d):4d} mean {d.mean():8.1f} median {d.median():8.1f} min {d.min():8.1f} max {d.max():8.1f}")
PY
find gpurun_out/r02k -name "*.csv" -size +2M -delete
