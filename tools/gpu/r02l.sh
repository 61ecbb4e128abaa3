mkdir -p gpurun_out/r02l
CHMC_PAR_SCAN=1 python tools/par_scan_stats.py 256 200 0.25 | cut -c1-400
for P in 0 1; do
CHMC_PAR_SCAN=$P python bench.py --config sir --no-cpu-baseline > gpurun_out/r02l/bench_sir_par$P.json 2>/dev/null
python - $P <<'PY'
import json,sys
P=sys.argv[1]
d=json.loads(open(f'gpurun_out/r02l/bench_sir_par{P}.json').read().strip().splitlines()[-1])
print('PAR_SCAN',P,'steps/s',round(d['value']),'ms',round(d['ms_per_step'],2),'succ',d['config']['step_success_rate'],'k',round(d['config']['mean_newton_iters_fwd_plus_bwd'],3), 'constr', d['config']['kernel_classes_warmup']['constr'])
PY
done
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir or config or full_size" 2>&1 | tail -3
