export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02v
cd /tmp && CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_z.so rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r02v/sq1 -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 4 --warmup 2 --burn-iters 2 > $R/gpurun_out/r02v/sq1.log 2>&1
cd /tmp && CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_z.so rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/r02v/sq2 -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 4 --warmup 2 --burn-iters 2 > $R/gpurun_out/r02v/sq2.log 2>&1
cd $R
python - <<'PY'
import pandas as pd, glob
for d in ('sq1','sq2'):
    fs=glob.glob(f'gpurun_out/r02v/{d}/**/*counter_collection.csv',recursive=True)
    if not fs: print(d,'no counters'); continue
    df=pd.read_csv(fs[0])
    df['k']=df['Kernel_Name'].str.replace(r'\(.*','',regex=True).str.replace('void ','').str.replace('chmc::','').str.slice(0,48)
    g=df.groupby(['k','Counter_Name'])['Counter_Value'].agg(['mean','count']).reset_index()
    for k in ('k_newton_lean<FhnModel, 7>','k_rev_wave<FhnModel, 7, 0, true>','k_gld_bwd_wave<FhnModel, 7>','k_gld_fwd_wave<FhnModel, 7>','k_fwd_scan<FhnModel, 7, true>'):
        sub=g[g.k==k]
        if len(sub): print(k, {r.Counter_Name: round(r['mean']) for _,r in sub.iterrows()}, 'n', int(sub['count'].iloc[0]))
PY
tail -3 gpurun_out/r02v/sq2.log | cut -c1-300
find gpurun_out/r02v -name "*.csv" -size +3M -delete
