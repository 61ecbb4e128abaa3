set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r02d
scan_stats() { python - "$1" <<'PY'
import pandas as pd, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
k=pd.read_csv(f).sort_values('Start_Timestamp')
k=k.iloc[int(len(k)*0.6):]
k['dur']=(k['End_Timestamp']-k['Start_Timestamp'])/1e3
for pat in ('k_fwd_scan','k_rev_wave<chmc::FhnModel, 7, 1','KUpdate<7, 0','KNewtonFactor','k_solve_chain_wave<chmc::FhnModel, 7, 0','KSymBlk'):
    d=k[k.Kernel_Name.str.contains(pat,regex=False)]['dur']
    if len(d): print(f"  {pat:45s} n {len(d):4d} mean {d.mean():7.1f} median {d.median():7.1f} p90 {d.quantile(0.9):7.1f}")
PY
}
# E1: quasi-Newton (scans without trajectory stores) with halves
cd /tmp && CHMC_HALVES=2 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02d/t_qn_h2 -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 6 --warmup 2 --burn-iters 2 --solver quasi-newton > $R/gpurun_out/r02d/t_qn_h2.log 2>&1; cd $R
echo "E1 qn halves"; scan_stats gpurun_out/r02d/t_qn_h2
cd /tmp && CHMC_HALVES=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02d/t_qn_h1 -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 6 --warmup 2 --burn-iters 2 --solver quasi-newton > $R/gpurun_out/r02d/t_qn_h1.log 2>&1; cd $R
echo "E1 qn single"; scan_stats gpurun_out/r02d/t_qn_h1
# E2: depth 6
cd /tmp && CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_depth6.so CHMC_HALVES=2 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02d/t_d6_h2 -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 6 --warmup 2 --burn-iters 2 > $R/gpurun_out/r02d/t_d6_h2.log 2>&1; cd $R
echo "E2 depth6 halves"; scan_stats gpurun_out/r02d/t_d6_h2
find gpurun_out/r02d -name "*.csv" -delete
for v in "1 0 main" "2 0 main" "2 1 main" "1 0 depth6" "2 0 depth6"; do set -- $v
  lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip.so; [ $3 = depth6 ] && lib=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_depth6.so
  if [ $2 = 1 ]; then export CHMC_NO_STAGGER=1; else unset CHMC_NO_STAGGER; fi
  CHMC_HIP_LIBRARY=$lib CHMC_HALVES=$1 python bench.py --no-cpu-baseline --no-profile > gpurun_out/r02d/b_$1_$2_$3.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r02d/b_$1_$2_$3.json').read().strip().splitlines()[-1]);print('halves $1 nostagger $2 lib $3:',round(d['value']),round(d['ms_per_step'],3))"
done
