python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_surface.py -m gpu -x -q -k "not posterior" 2>&1 | tail -4
for i in 1 2; do for G in 1 0; do CHMC_XOBS_PAR=$G python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config']['kernel_classes_warmup']; print('xobs_par', $G, round(d['value']), round(d['ms_per_step'],3), 'other', t['other'], 'ok', d['config']['step_success_rate'], d['config']['mean_newton_iters_fwd_plus_bwd'])"; done; done
