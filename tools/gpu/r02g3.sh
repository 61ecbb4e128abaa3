python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_surface.py -m gpu -x -q -k "not posterior" 2>&1 | tail -4
for G in 1 0; do for i in 1 2; do CHMC_STATE_LEAN=$G python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config']['kernel_classes_warmup']; print('state_lean', $G, round(d['value']), round(d['ms_per_step'],3), 'state', t['state_blk']['ms_per_step'], 'ok', d['config']['step_success_rate'], d['config']['mean_newton_iters_fwd_plus_bwd'])"; done; done
