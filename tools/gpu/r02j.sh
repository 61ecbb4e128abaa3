R=$PWD
CHMC_PAR_SCAN=1 python tools/par_scan_stats.py 64 200 0.25
CHMC_PAR_SCAN=1 python tools/par_scan_stats.py 64 200 0.05
for lib in A; do
 echo "== variant $lib (k_gld_bwd_wave<16> fully unrolled)"
 CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_$lib.so CHMC_PAR_SCAN=0 python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -q -k "sir" 2>&1 | tail -3
 CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/libchmc_hip_$lib.so CHMC_PAR_SCAN=0 python bench.py --config sir --no-cpu-baseline > gpurun_out/bench_sir_$lib.json 2>/dev/null
 python -c "
import json;d=json.loads(open('gpurun_out/bench_sir_$lib.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2),d['config']['kernel_classes_warmup']['grad_log_det_blk'])"
done
