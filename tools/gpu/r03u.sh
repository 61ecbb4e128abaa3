export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "row_split or halves or sir or mfma" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
