# the end-to-end examples on the build in the tree (logs under gpurun_out/<tag>/): SIR boarding school (static trajectories,
# per-chain kernels), FHN static trajectories, FHN no-U-turn trees with per-chain and with one shared step size
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04v}; mkdir -p $O; rm -rf $O/*
timeout -k 10 300 python examples/sir_boarding_school_chmc.py 256 300 100 16 > $O/example_sir_boarding_school_256.log 2>&1; tail -4 $O/example_sir_boarding_school_256.log
timeout -k 10 400 python examples/fhn_noisy_chmc.py 256 400 120 40 > $O/example_fhn_sampler_256x400.log 2>&1; tail -4 $O/example_fhn_sampler_256x400.log
timeout -k 10 400 python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic > $O/example_fhn_dynamic_256x400.log 2>&1; tail -4 $O/example_fhn_dynamic_256x400.log
timeout -k 10 400 python examples/fhn_noisy_chmc.py 256 400 60 25 - dynamic-shared > $O/example_fhn_dynamic_shared_step_256x400.log 2>&1; tail -4 $O/example_fhn_dynamic_shared_step_256x400.log
