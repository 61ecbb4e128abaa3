R=$PWD
mkdir -p gpurun_out/r02i
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir or config or full_size or half" > gpurun_out/r02i/pytest_par.log 2>&1; tail -4 gpurun_out/r02i/pytest_par.log
for P in 0 1; do
CHMC_PAR_SCAN=$P python bench.py --config sir --no-cpu-baseline > gpurun_out/r02i/bench_sir_par$P.json 2>/dev/null
python - $P <<'PY'
import json,sys
P=sys.argv[1]
d=json.loads(open(f'gpurun_out/r02i/bench_sir_par{P}.json').read().strip().splitlines()[-1])
print('PAR_SCAN',P,'steps/s',round(d['value']),'ms',round(d['ms_per_step'],2),'succ',d['config']['step_success_rate'],'k',round(d['config']['mean_newton_iters_fwd_plus_bwd'],3))
for k,v in d['config']['kernel_classes_warmup'].items(): print('   ',k,v['ms_per_step'],v['ms_per_launch'],v['launches_per_step'])
PY
done
python - <<'PY'
# fallback counter on the SIR workload
import numpy as np, sys
sys.path.insert(0,'.')
from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
wl=SirWorkload(64,num_steps_per_obs=200)
wl.refresh_momentum()
for i in range(20): r=wl.step(0.25)
print('counters',wl.ctx.counters(), 'ok', (r['status']==0).mean())
PY
for lib in A B C; do
 echo "== variant $lib"
 CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_$lib.so python -m pytest tests/test_hip_parity.py -m gpu -q -k "ops_small and sir-14" 2>&1 | grep -E "grad_log_det|passed|failed" | cut -c1-400 | tail -3
done
