export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "time_parallel" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
