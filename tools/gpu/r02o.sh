mkdir -p gpurun_out/r02o
python tools/par_scan_compare.py 256 200 0.25 8 | cut -c1-500
for dt in 0.25 0.15; do for P in 0 1; do
CHMC_PAR_SCAN=$P python bench.py --config sir --no-cpu-baseline --step-size $dt > gpurun_out/r02o/bench_sir_par${P}_$dt.json 2>/dev/null
python - $P $dt <<'PY'
import json,sys
P,dt=sys.argv[1:]
d=json.loads(open(f'gpurun_out/r02o/bench_sir_par{P}_{dt}.json').read().strip().splitlines()[-1])
print('dt',dt,'PAR_SCAN',P,'steps/s',round(d['value']),'ms',round(d['ms_per_step'],2),'succ',round(d['config']['step_success_rate'],4),'k',round(d['config']['mean_newton_iters_fwd_plus_bwd'],3), 'constr ms/launch', d['config']['kernel_classes_warmup']['constr']['ms_per_launch'])
PY
done; done
