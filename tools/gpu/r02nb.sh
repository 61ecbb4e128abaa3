mkdir -p gpurun_out/r02nb
python -m pytest tests/test_notebook_posterior.py -m gpu -x -q 2>&1 | tail -3
python examples/fhn_notebook_posterior.py 64 700 200 24 > gpurun_out/r02nb/notebook_posterior_64x700.log 2>&1; tail -14 gpurun_out/r02nb/notebook_posterior_64x700.log
python examples/fhn_notebook_posterior.py 64 450 150 0 - dynamic > gpurun_out/r02nb/notebook_posterior_dynamic_64x450.log 2>&1; tail -12 gpurun_out/r02nb/notebook_posterior_dynamic_64x450.log
python bench.py > gpurun_out/r02nb/bench_fhn_noisy.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r02nb/bench_fhn_noisy.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), d['ms_per_step'], r['kernel'], r['bound'], r['frac'], r['traffic'], r['hbm_traffic_frac'], r['traffic_source'][:40])"
