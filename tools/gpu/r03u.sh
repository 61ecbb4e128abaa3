export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "time_parallel or row_split" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
