export TMPDIR=/tmp
O=$PWD/gpurun_out/r04e; mkdir -p $O
for h in 1 2 1 2; do CHMC_HALVES=$h timeout -k 10 300 python bench.py --no-cpu-baseline 2> $O/e.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('halves $h', round(d['value']), round(d['ms_per_step'], 3), {k: round(v['ms_per_step'], 3) for k, v in c['kernel_classes_warmup'].items() if isinstance(v, dict) and 'ms_per_step' in v})
"; done
for h in 1 2; do CHMC_HALVES=$h timeout -k 10 300 python bench.py --no-cpu-baseline --chains-per-gpu 512 --num-steps-per-obs 800 2> $O/e.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('S800x512 halves $h', round(d['value']), round(d['ms_per_step'], 3))
"; done
