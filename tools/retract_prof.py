"""Phase breakdown of the per-chain retraction kernel (k_retract_chain) on the boarding-school SIR workload.

Needs a diagnostic build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCHMC_RETRACT_PROF -o build/libchmc_prof.so csrc/chmc.hip
    CHMC_HIP_LIBRARY=build/libchmc_prof.so python tools/retract_prof.py [chains] [trajectories]
"""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from manifold_mcmc_for_diffusions_amd.workload import SirWorkload  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ntraj = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = SirWorkload(B, num_steps_per_obs=200)
ctx = wl.ctx
for _ in range(2):  # burn-in as bench.py does
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    for _ in range(16):
        r = wl.step(0.25, active=act)
        act &= (r["status"] == 0).astype(np.int32)
d0 = ctx.diagnostics()["par_scan"].copy()
x0 = ctx.diagnostics()["extra"].copy()
t0 = time.perf_counter()
its, steps = [], 0
for _ in range(ntraj):
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    r = ctx.leapfrog_steps(0.25, 16, active=act, **wl.solver)
    steps += int((r["n_done"] + (r["status"] > 0)).sum())
    its.append((r["iters_fwd"] + r["iters_bwd"]).sum())
el = time.perf_counter() - t0
d = ctx.diagnostics()["par_scan"] - d0
ticks = d[48:52].astype(float) * 0.01  # us (100 MHz)
n_it, n_ret = int(d[52]), int(d[53])
sweeps = d[1:48]
print(f"{B} chains, {ntraj} trajectories: {steps} chain-steps in {el*1e3:.1f} ms = {steps/el:.0f} steps/s")
print(f"retractions (chain level) {n_ret}, iterations {n_it} ({n_it/max(n_ret,1):.2f} per retraction)")
if n_it:
    names = ("scan", "interval sums", "combine + factor", "update + check")
    for nm, t in zip(names, ticks):
        print(f"  {nm:18s} {t/n_it:8.1f} us per iteration  ({100*t/ticks.sum():.0f} %)")
    print(f"  total              {ticks.sum()/n_it:8.1f} us per iteration")
sw = d[56:63].astype(float) * 0.01
nsw = float((d[1:48] * (np.arange(47) % 16 + 1)).sum() + d[1:48].sum())  # sweeps incl. the final passes
if sw.sum() > 0:
    for nm, t in zip(("recursion", "in-wave scan", "cross-wave", "new start states", "absorbing fronts", "hand-over", "final pass"), sw):
        print(f"    sweep: {nm:18s} {t/n_it:8.2f} us per iteration ({100*t/sw.sum():.0f} %)")
xt = (ctx.diagnostics()["extra"] - x0).astype(float) * 0.16  # (ticks >> 4 at 100 MHz)
if xt.sum() > 0:
    names = ("A(dt/2) + flow", "both retractions", "state scan", "interval sums (state)", "combine / Cholesky / prep / prologue",
             "grad log det forward", "grad log det backward (2 phases)", "finish + chain", "momentum fix", "J p + core solves",
             "mu_F + J^T lambda + reverse flow", "reverse check + A(dt/2) + commit")
    print(f"k_traj_chain phases, us per chain-step ({steps} chain-steps):")
    for nm, t in zip(names, xt):
        print(f"    {nm:40s} {t/steps:8.1f}  ({100*t/xt.sum():.0f} %)")
    print(f"    {'total':40s} {xt.sum()/steps:8.1f}")
if os.environ.get("CHMC_COMB_PROF_BUILD"):
    print(f"    combine of a Newton iteration: steps (a)-(e) {d[54]*0.01/n_it:.1f} us, (f) LU / core / multipliers / mu_F {d[55]*0.01/n_it:.1f} us")
elif d[54] + d[55] + d[63] > 0:
    print(f"    inside 'combine / Cholesky / prep / prologue' (wavefront 0): combine + Cholesky {d[54]*0.01/steps:.1f}, chain core "
          f"{d[55]*0.01/steps:.1f}, grad-log-det preparation {d[63]*0.01/steps:.1f} us per chain-step; the rest is the interval prologue")
ns = sweeps[:15] + sweeps[16:31] + sweeps[32:47]
tot = ns.sum()
print("sweeps to settle (histogram over scans, +1 final sweep each):", ns.tolist(), "mean", (ns * np.arange(1, 16)).sum() / max(tot, 1))
