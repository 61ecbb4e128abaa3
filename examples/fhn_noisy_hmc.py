#!/usr/bin/env python3
"""The reference's unconstrained-HMC comparator (scripts/fhn_model_noisy_obs_hmc_experiment.py) on one MI355X: batched
HMC on conditioned_diffusion_neg_log_dens_and_grad with the observation noise marginalised, same data and initial
states as examples/fhn_noisy_chmc.py.   usage: fhn_noisy_hmc.py [chains] [S] [iters] [warm-up] [identity|diagonal|block]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload  # noqa: E402
from manifold_mcmc_for_diffusions_amd.hmc import sample_hmc  # noqa: E402
from manifold_mcmc_for_diffusions_amd import example_models as em  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 150
n_warm = int(sys.argv[4]) if len(sys.argv) > 4 else 50
metric_type = sys.argv[5] if len(sys.argv) > 5 else "identity"
n_step = 16
wl = FhnWorkload(B, num_steps_per_obs=S, device_init=True)
ctx = wl.ctx
q0 = ctx.get_state(want_p=False)[0][:, :ctx.U + ctx.NV]
t0 = time.time()
res = sample_hmc(ctx, q0, n_iter, n_step, 0.01, seed=wl.seed, n_adapt=n_warm, metric_type=metric_type,
                 callback=lambda it, h, a, e: (it % 10 == 0) and print(
                     f"  iter {it:4d} accept {a:.2f} step {e:.4f} z-median {np.median(em.fhn.generate_z(h[:, :4]), 0).round(3)}",
                     flush=True))
el = time.time() - t0
z = em.fhn.generate_z(res["heads"][n_warm:, :, :4])
print(f"{n_iter} transitions x {n_step} leapfrog steps x {B} chains in {el:.1f} s = {n_iter * n_step * B / el:.0f} "
      f"unconstrained leapfrog steps/s ({metric_type} metric); final step size {res['final_step_size']:.4f}, "
      f"mean accept (main) {res['accept_stat'][n_warm:].mean():.3f}")
for k, nm in enumerate(("sigma", "epsilon", "gamma", "beta")):
    v = z[:, :, k].ravel()
    print(f"  {nm:8s} true {(0.3, 0.1, 1.5, 0.8)[k]:.2f}  median {np.median(v):.3f}  5%-95% [{np.quantile(v, 0.05):.3f}, {np.quantile(v, 0.95):.3f}]")
print("(the stiff noise-marginalised target forces step sizes two orders of magnitude below the constrained sampler's: the"
      " reference's motivation for the manifold method; compare examples/fhn_noisy_chmc.py)")
