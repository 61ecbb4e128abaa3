mkdir -p gpurun_out/r02x
python -m pytest tests -m gpu -x -q -k "tree or dynamic or surface or steps_small" > gpurun_out/r02x/pytest.log 2>&1; tail -3 gpurun_out/r02x/pytest.log
python tools/dynamic_timing.py > gpurun_out/r02x/dynamic_timing.log 2>&1; cat gpurun_out/r02x/dynamic_timing.log
python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic > gpurun_out/r02x/example_fhn_dynamic_256x400.log 2>&1; tail -9 gpurun_out/r02x/example_fhn_dynamic_256x400.log
