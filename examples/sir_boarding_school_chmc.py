#!/usr/bin/env python3
"""End-to-end example on an MI355X: the SIR model with a time-varying contact rate on the boarding-school influenza
data, as scripts/sir_model_chmc_experiment.py of the reference sets it up (14 daily counts, 20 steps per observation,
one sub-sequence of 14 observations, sigma_y = 1): batched initial states by the Adam-based finder of the noisy system
(sde/mici_extensions.py:1679-1801), then batched constrained HMC.
usage: sir_boarding_school_chmc.py [chains] [iterations] [warm-up] [steps per trajectory] [output dir]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd import example_models as em, init  # noqa: E402
from manifold_mcmc_for_diffusions_amd.context import ChmcContext  # noqa: E402
from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc  # noqa: E402

a = sys.argv[1:]
B = int(a[0]) if len(a) > 0 else 256
n_iter = int(a[1]) if len(a) > 1 else 300
n_warm = int(a[2]) if len(a) > 2 else 100
n_step = int(a[3]) if len(a) > 3 else 16
out_dir = a[4] if len(a) > 4 else None
data = np.load(os.path.join(ROOT, "tests", "golden", "reference_data", "sir_model_boarding_school_data.npz"))
y, obs_interval = np.asarray(data["y_seq"], dtype=np.float64).reshape(-1), float(data["obs_interval"])
m = em.sir
ctx = ChmcContext("sir", obs_interval, 20, 14, y, sigma=1.0, num_chains=B)
print(f"SIR: {B} chains, dim_q = {ctx.Q}, {ctx.num_blocks} sub-sequence of {len(y)} observations, {ctx.RM}-row kernels")
rng = np.random.default_rng(20200710)
t0 = time.time()
q, xo, tries = init.find_initial_states_by_gradient_descent_noisy_system(ctx, rng, adam_step_size=1e-1, max_iters=5000,
                                                                         log=print)
print(f"initial states in {time.time() - t0:.1f} s (restarts per chain: max {tries.max() - 1}), |c|max "
      f"{np.abs(ctx.constr()).max():.1e}")


def trace_func(head, ham):  # scripts/sir_model_chmc_experiment.py:75-94
    z = m.generate_z(head[:, :4])
    return {"α₀": m.generate_x_0(z, head[:, 4:5])[:, -1], "β": z[:, 0], "γ": z[:, 1], "ζ": z[:, 2], "ϵ": z[:, 3],
            "hamiltonian": ham}


t0 = time.time()
res = sample_static_chmc(ctx, n_iter, n_step, 0.05, seed=7, n_adapt=n_warm, n_head=5, jitter_length=True,
                         trace_dir=out_dir or os.path.join(ROOT, "gpurun_out", "sir_run"), trace_func=trace_func,
                         callback=lambda it, h, acc, e: (it % 25 == 0) and print(
                             f"  iter {it:4d} accept {acc:.2f} step {e:.3f}", flush=True))
el = time.time() - t0
sm = res["summary"]
print(f"{n_iter} transitions x <= {n_step} steps x {B} chains in {el:.1f} s; final step size "
      f"{res['final_step_size']:.3f}, accept {res['accept_stat'][n_warm:].mean():.2f}, failed trajectories "
      f"{res['fail_rate'][n_warm:].mean():.3f}")
for k in ("α₀", "β", "γ", "ζ", "ϵ"):
    print(f"  {k:3s} mean {sm['mean'][k]:8.4f} sd {sm['sd'][k]:7.4f}  r_hat {sm['r_hat'][k]:.3f} ess {sm['ess_bulk'][k]:7.0f}")
