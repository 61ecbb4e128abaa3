# the whole GPU suite, then the default bench line without the CPU baseline (quick look at the headline)
export TMPDIR=/tmp
TAG=${1:-suite}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_default.json 2> $O/e.log || tail -5 $O/e.log
python - $O/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d['config']
print(round(d['value']), round(d['ms_per_step'], 3), c['value_repeats']['values'], {k: v['ms_per_step'] for k, v in c['kernel_classes_warmup'].items()})
PY
