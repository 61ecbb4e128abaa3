python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "compact_row_kernels" 2>&1 | tail -15
