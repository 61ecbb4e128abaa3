mkdir -p gpurun_out/r02f
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "inner" > gpurun_out/r02f/pytest.log 2>&1; tail -5 gpurun_out/r02f/pytest.log
