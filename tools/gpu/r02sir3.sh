mkdir -p gpurun_out/r02sir3
python -m pytest tests/test_hip_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
for n in 256 1024; do python bench.py --config sir --chains-per-gpu $n --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sir', $n, round(d['value']), round(d['ms_per_step'],3), d['config']['step_success_rate'])"; done
python tools/sir_timing.py 1024 200 14 2>&1 | tail -2
