mkdir -p gpurun_out/r02q
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir" 2>&1 | tail -3
python tools/par_scan_compare.py 256 200 0.25 6 | cut -c1-400
python bench.py --config sir --no-cpu-baseline > gpurun_out/r02q/bench_sir.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02q/bench_sir.json').read().strip().splitlines()[-1])
print('steps/s',round(d['value']),'ms',round(d['ms_per_step'],2),'succ',round(d['config']['step_success_rate'],4),'k',round(d['config']['mean_newton_iters_fwd_plus_bwd'],3))
for k,v in d['config']['kernel_classes_warmup'].items(): print('   ',k,v['ms_per_step'],v['ms_per_launch'],v['launches_per_step'])
PY
