mkdir -p gpurun_out/r02s2
python -m pytest tests -m gpu -x -q > gpurun_out/r02s2/pytest.log 2>&1; tail -4 gpurun_out/r02s2/pytest.log
python tools/sir_timing.py 1024 200 2 2>&1 | tail -2
python bench.py --no-cpu-baseline --config fhn_noiseless 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noiseless', round(d['value']), round(d['ms_per_step'],3))"
python bench.py --no-cpu-baseline --solver quasi-newton 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qn', round(d['value']), round(d['ms_per_step'],3))"
