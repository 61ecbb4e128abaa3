mkdir -p gpurun_out/r02nb2
python examples/fhn_notebook_posterior.py 64 450 150 0 - dynamic > gpurun_out/r02nb2/notebook_posterior_dynamic_64x450.log 2>&1; tail -12 gpurun_out/r02nb2/notebook_posterior_dynamic_64x450.log
python tools/dynamic_timing.py > gpurun_out/r02nb2/dynamic_timing.log 2>&1; head -3 gpurun_out/r02nb2/dynamic_timing.log | tail -2
python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic > gpurun_out/r02nb2/example_fhn_dynamic_256x400.log 2>&1; grep "leapfrog steps/s" gpurun_out/r02nb2/example_fhn_dynamic_256x400.log
