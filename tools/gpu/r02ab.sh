# A/B of library variants: bash tools/gpu/r02ab.sh lib1.so lib2.so ...   (two bench runs each, interleaved)
R=$PWD
mkdir -p gpurun_out/r02ab
for i in 1 2; do for lib in "$@"; do
CHMC_HIP_LIBRARY=$R/manifold_mcmc_for_diffusions_amd/$lib python bench.py --no-cpu-baseline > gpurun_out/r02ab/b_$lib.$i.json 2>/dev/null
python - $lib $i <<'PY'
import json,sys
lib,i=sys.argv[1:]
d=json.loads(open(f'gpurun_out/r02ab/b_{lib}.{i}.json').read().strip().splitlines()[-1])
t=d['config']['kernel_classes_warmup']
print(lib,i,round(d['value']),round(d['ms_per_step'],3),'gld',t['grad_log_det_blk']['ms_per_step'],'state',t['state_blk']['ms_per_step'],'newton',t['newton_blk']['ms_per_step'],'ok',d['config']['step_success_rate'])
PY
done; done
