mkdir -p gpurun_out/r02g4
for i in 1 2 3; do for G in 1 0; do CHMC_STATE_LEAN=$G python bench.py --no-cpu-baseline --steps 64 2>/dev/null > gpurun_out/r02g4/b_$G.$i.json; python -c "
import json; d=json.loads(open('gpurun_out/r02g4/b_$G.$i.json').read().strip().splitlines()[-1]); t=d['config']['kernel_classes_warmup']; print('state_lean', $G, round(d['value']), round(d['ms_per_step'],3), ' '.join(k+':'+str(v['ms_per_step']) for k,v in sorted(t.items(), key=lambda kv:-kv[1]['ms_per_step'])))"; done; done
