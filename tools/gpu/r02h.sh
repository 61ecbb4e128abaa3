export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02h
CHMC_GRAM_MFMA=1 python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir" > gpurun_out/r02h/pytest_mfma.log 2>&1; tail -3 gpurun_out/r02h/pytest_mfma.log
for G in 0 1; do
  cd /tmp && CHMC_GRAM_MFMA=$G rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02h/sir_g$G -- python3 $R/bench.py --config sir --no-cpu-baseline --no-profile --steps 16 --warmup 2 > $R/gpurun_out/r02h/sir_g$G.log 2>&1; cd $R
  python - $G <<'PY'
import pandas as pd, glob, sys, json
G=sys.argv[1]
f=glob.glob(f'gpurun_out/r02h/sir_g{G}/**/*kernel_stats.csv',recursive=True)[0]
k=pd.read_csv(f)
k['name']=k['Name'].str.replace(r'\(.*','',regex=True).str.replace('void ','').str.replace('chmc::','').str.slice(0,60)
print(f"--- CHMC_GRAM_MFMA={G}")
print(k[['name','Calls','TotalDurationNs','AverageNs','Percentage']].head(14).to_string())
try:
    d=json.loads([l for l in open(f'gpurun_out/r02h/sir_g{G}.log') if l.startswith('{')][-1]); print('steps/s',round(d['value']),'ms',round(d['ms_per_step'],2))
except Exception as e: print(e)
PY
  cp $(find gpurun_out/r02h/sir_g$G -name "*kernel_stats.csv") gpurun_out/r02h/sir_kernel_stats_mfma$G.csv
  find gpurun_out/r02h/sir_g$G -name "*.csv" -size +5M -delete
done
