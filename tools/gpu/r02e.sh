R=$PWD
mkdir -p gpurun_out/r02e
CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_nb4p.so python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "forward_scan or half or config2" > gpurun_out/r02e/pytest_nb4p.log 2>&1; tail -3 gpurun_out/r02e/pytest_nb4p.log
for lib in nb1 nb4 nb4p prio; do for H in 1 2; do
  CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_$lib.so CHMC_HALVES=$H python bench.py --no-cpu-baseline --no-profile > gpurun_out/r02e/b_${lib}_$H.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r02e/b_${lib}_$H.json').read().strip().splitlines()[-1]);print('lib $lib halves $H:',round(d['value']),round(d['ms_per_step'],3))"
done; done
