#!/usr/bin/env python3
"""End-to-end example on an MI355X: posterior sampling for the FitzHugh-Nagumo model with noisy observations
(the configuration of scripts/fhn_model_noisy_obs_chmc_experiment.py in the reference: T = 100 observations,
R = 5, sigma_y = 0.1) with batched constrained HMC.   usage: fhn_noisy_chmc.py [chains] [S] [iters] [warm-up] [output dir] [static|dynamic|dynamic-shared|metric]
`metric`: static trajectories with the block-diagonal metric adapter of the reference (sde/mici_extensions.py:1804-1931) on the
four global parameters during the warm-up.
With an output directory the traced variables of the reference's trace function (sigma, epsilon, gamma, beta, x_0,
hamiltonian) are written as memory-mapped `.npy` files together with `summary.json`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload  # noqa: E402
from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc  # noqa: E402
from manifold_mcmc_for_diffusions_amd import example_models as em  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 150
n_warm = int(sys.argv[4]) if len(sys.argv) > 4 else 50
out_dir = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
dynamic = len(sys.argv) > 6 and sys.argv[6] in ("dynamic", "dynamic-shared")  # the reference's transition (no-U-turn trees) instead of 16 fixed steps
shared_step = len(sys.argv) > 6 and sys.argv[6] == "dynamic-shared"        # one shared step size in the warm-up instead of one per chain
adapt_metric = len(sys.argv) > 6 and sys.argv[6] == "metric"


def trace_func(head, ham):  # scripts/fhn_model_noisy_obs_chmc_experiment.py:82-99
    z = em.fhn.generate_z(head[:, :4])
    return {"σ": z[:, 0], "ϵ": z[:, 1], "γ": z[:, 2], "β": z[:, 3], "x_0": em.fhn.generate_x_0(z, head[:, 4:6]),
            "hamiltonian": ham}


t0 = time.time()
wl = FhnWorkload(B, num_steps_per_obs=S, device_init=True)  # initial states solved on the device
print(f"set-up {time.time() - t0:.1f} s: {B} chains, dim_q = {wl.ctx.Q}", flush=True)
t0 = time.time()
if dynamic:
    from manifold_mcmc_for_diffusions_amd.dynamic import sample_dynamic_chmc  # noqa: E402
    res = sample_dynamic_chmc(wl.ctx, n_iter, 0.1, seed=wl.seed, n_adapt=n_warm, trace_dir=out_dir, trace_func=trace_func,
                              per_chain_step_size=not shared_step,
                              callback=lambda it, h, a, e, st: (it % 10 == 0) and print(
                                  f"  iter {it:4d} accept {a:.2f} step {e:.3f} steps per tree {st['n_step'].mean():.1f} "
                                  f"z-median {np.median(em.fhn.generate_z(h[:, :4]), 0).round(3)}", flush=True))
    res["fail_rate"] = res["integrator_error"]
    steps_total = float(res["n_step"].sum()) * B
else:
    from manifold_mcmc_for_diffusions_amd.adapters import OnlineBlockDiagonalMetricAdapter  # noqa: E402
    res = sample_static_chmc(wl.ctx, n_iter, 16, 0.1, seed=wl.seed, n_adapt=n_warm, trace_dir=out_dir, trace_func=trace_func,
                             metric_adapter=OnlineBlockDiagonalMetricAdapter(4) if adapt_metric else None,
                             callback=lambda it, h, a, e: (it % 10 == 0) and print(
                                 f"  iter {it:4d} accept {a:.2f} step {e:.3f} z-median {np.median(em.fhn.generate_z(h[:, :4]), 0).round(3)}",
                                 flush=True))
    steps_total = n_iter * 16 * B
el = time.time() - t0
z = em.fhn.generate_z(res["heads"][n_warm:, :, :4])  # [iters, B, 4] = sigma, eps, gamma, beta
x0 = em.fhn.generate_x_0(z, res["heads"][n_warm:, :, 4:6])
print(f"{n_iter} transitions x {'dynamic trees' if dynamic else '16 steps'} x {B} chains in {el:.1f} s = {steps_total / el:.0f} leapfrog steps/s "
      f"(includes momentum refresh, accept/reject, partition switch, host traces)")
if adapt_metric:
    print("adapted M_0 (inverse of the regularised covariance of u over all chains):\n", np.round(res["metric_M_0"], 2))
print("final step size", round(res["final_step_size"], 4), "mean accept (main)", res["accept_stat"][n_warm:].mean().round(3),
      "failed trajectories", res["fail_rate"][n_warm:].mean().round(4))
names = ["sigma", "epsilon", "gamma", "beta"]
truth = [0.3, 0.1, 1.5, 0.8]
# A prior draw far out in the tails (initial Hamiltonian 1e8 and more) needs longer than a short warm-up to travel to the
# bulk, or is released only late from a stiff start: chains whose parameters at the first main transition are still
# further than 8 robust standard deviations from the batch median, or that move in fewer than half of the main
# transitions, are counted and left out of the summary below (summary.json keeps every chain).
u0 = np.log(np.abs(z[0])) if z.shape[0] else np.zeros((B, 4))
med = np.median(u0, 0)
rsd = 1.4826 * np.median(np.abs(u0 - med), 0) + 1e-12
settled = (np.abs(u0 - med) < 8 * rsd).all(1) & ((np.diff(z[:, :, 0], axis=0) != 0).mean(0) >= 0.5)
print(f"  {int(settled.sum())} of {B} chains settled by the end of the warm-up (summary over those); left out: "
      f"{np.flatnonzero(~settled).tolist()}")
z_all, z = z, z[:, settled]
x0 = x0[:, settled]
chain_means = z.mean(0)  # [B, 4]
for k in range(4):
    allv = z[:, :, k].ravel()
    between = chain_means[:, k].var(ddof=1)
    within = z[:, :, k].var(0, ddof=1).mean()
    n = z.shape[0]
    rhat = np.sqrt(((n - 1) / n * within + between) / within)
    print(f"  {names[k]:8s} true {truth[k]:.2f}  posterior mean {allv.mean():.3f} sd {allv.std():.3f}  "
          f"5%-95% [{np.quantile(allv, 0.05):.3f}, {np.quantile(allv, 0.95):.3f}]  split-free R-hat {rhat:.3f}")
print(f"  x_0      true [-0.5, 0.2]  posterior mean {x0.reshape(-1, 2).mean(0).round(3)}")
if out_dir:
    sm = res["summary"]
    print(f"traces and summary.json in {out_dir}: " + ", ".join(
        f"{k} {sm['mean'][k]:.3f} (r_hat {sm['r_hat'][k]:.3f}, ess {sm['ess_bulk'][k]:.0f})" for k in ("σ", "ϵ", "γ", "β")))
