export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02sir
mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --config sir --no-cpu-baseline --no-profile --steps 6 --warmup 2 > $O/trace.log 2>&1; cd $R
python tools/trace_overlap.py $O/trace 0.3 > $O/overlap.txt 2>&1
python - <<'PY' > gpurun_out/r02sir/timeline.txt 2>&1
import glob, pandas as pd
k = pd.read_csv(glob.glob('gpurun_out/r02sir/trace/**/*kernel_trace.csv', recursive=True)[0]).sort_values('Start_Timestamp').reset_index(drop=True)
k['name'] = k['Kernel_Name'].str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.replace('chmc::', '').str.slice(0, 48)
n = len(k)
seg = k.iloc[int(n * 0.80):int(n * 0.80) + 160]
t0 = seg['Start_Timestamp'].iloc[0]
scol = 'Stream_Id' if 'Stream_Id' in k.columns else 'Queue_Id'
for _, r in seg.iterrows():
    print(f"{(r['Start_Timestamp'] - t0) / 1e3:9.1f} us  +{(r['End_Timestamp'] - r['Start_Timestamp']) / 1e3:8.1f}  s{r[scol]}  grid {r.get('Grid_Size_X', '')}  {r['name']}")
PY
find $O/trace -name "*.csv" -size +20M -delete
cat $O/overlap.txt | head -30
