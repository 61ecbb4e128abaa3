mkdir -p gpurun_out/r02r
python -m pytest tests -m gpu -x -q > gpurun_out/r02r/pytest.log 2>&1; tail -5 gpurun_out/r02r/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/r02r/bench.json 2>/dev/null
python -c "
import json;d=json.loads(open('gpurun_out/r02r/bench.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],3), d['roofline']['frac'])"
