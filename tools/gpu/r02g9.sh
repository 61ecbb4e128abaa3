CHMC_TWO_PHASE8=1 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "steps_small or full_size_distinct" 2>&1 | tail -3
for i in 1 2; do for G in 1 0; do CHMC_TWO_PHASE8=$G python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config']['kernel_classes_warmup']; print('two_phase8', $G, round(d['value']), round(d['ms_per_step'],3), 'newton', t['newton_blk'], 'ok', d['config']['step_success_rate'], d['config']['mean_newton_iters_fwd_plus_bwd'])"; done; done
