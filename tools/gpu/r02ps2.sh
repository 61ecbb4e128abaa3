for P in 0 1; do CHMC_PAR_SCAN=$P python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config']['kernel_classes_warmup']; print('par_scan', $P, round(d['value']), round(d['ms_per_step'],3), 'constr', t['constr'], 'ok', d['config']['step_success_rate'])"; done
