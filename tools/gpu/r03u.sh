export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "adam or initial_states" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
