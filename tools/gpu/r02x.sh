mkdir -p gpurun_out/r02x
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/dynamic_timing.py > gpurun_out/r02x/dynamic_timing.log 2>&1; cat gpurun_out/r02x/dynamic_timing.log
python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic > gpurun_out/r02x/example_fhn_dynamic_256x400.log 2>&1; tail -9 gpurun_out/r02x/example_fhn_dynamic_256x400.log
