mkdir -p gpurun_out/r02y
for L in 1 0; do for i in 1 2; do echo "CHMC_NEWTON_LEAN=$L"; CHMC_NEWTON_LEAN=$L python tools/sir_timing.py 1024 200 2 2>&1 | tail -2; done; done
python -m pytest tests/test_notebook_posterior.py -m gpu -x -q 2>&1 | tail -3
