mkdir -p gpurun_out/r02sir2
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q -k "sir or SIR or boarding" 2>&1 | tail -4
for C in 1; do for i in 1 2; do
CHMC_COMPACT16=$C python bench.py --config sir --no-cpu-baseline 2>/dev/null > gpurun_out/r02sir2/b_$C.$i.json
python - $C $i <<'PY'
import json,sys
C,i=sys.argv[1:]
d=json.loads(open(f'gpurun_out/r02sir2/b_{C}.{i}.json').read().strip().splitlines()[-1])
t=d['config']['kernel_classes_warmup']
print('compact16',C,i,round(d['value']),round(d['ms_per_step'],3),'newton',t['newton_blk'],'constr',t['constr']['ms_per_step'],'ok',d['config']['step_success_rate'], d['config']['mean_newton_iters_fwd_plus_bwd'])
PY
done; done
