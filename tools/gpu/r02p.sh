python -m pytest tests/test_hip_surface.py -m gpu -x -q -k "gather" 2>&1 | tail -5
