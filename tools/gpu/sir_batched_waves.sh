export TMPDIR=/tmp
O=$PWD/gpurun_out/r04j; mkdir -p $O; rm -rf $O/*
for w in 2 4; do for b in 256 1024; do
CHMC_RETRACT_KERNEL=0 CHMC_PAR_WAVES=$w timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --config sir --chains-per-gpu $b 2> $O/e1.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('batched W=$w B=$b', round(d['value']), round(d['ms_per_step'], 3), 'rounds', c.get('newton_rounds_per_step'), 'launches', c.get('launches_per_step'), (c.get('value_repeats') or {}).get('values'))"
done; done
