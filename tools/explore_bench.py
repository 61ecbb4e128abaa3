"""Exploration helper (not part of the product): burn-in behaviour, step-size choice and per-kernel timings of the
FHN noisy S=400 workload on one MI355X.  usage: explore_bench.py B S burn_dt burn_iters burn_steps"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ctypes as C
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
from manifold_mcmc_for_diffusions_amd import _lib
B = int(sys.argv[1]); S = int(sys.argv[2]); bdt = float(sys.argv[3]); biters = int(sys.argv[4]); bsteps = int(sys.argv[5])
t0 = time.time()
wl = FhnWorkload(B, num_steps_per_obs=S)
ctx = wl.ctx
L = _lib.lib()
dev = torch.device("cuda:0")
print("setup", round(time.time() - t0, 1), "Q", ctx.Q, flush=True)
def traj(dt, n, mh=False):
    wl.refresh_momentum_device(torch, dev)
    h0 = ctx.hamiltonian()
    act = np.ones(B, dtype=np.int32); its = []
    for s in range(n):
        r = wl.step(dt, active=act)
        ok = r["status"] == 0
        its.append(((r["iters_fwd"][ok] + r["iters_bwd"][ok]).mean() if ok.any() else 0))
        act = act & ok
    h1 = ctx.hamiltonian()
    return h0, h1, act, np.mean(its)
for it in range(biters):
    t1 = time.time()
    h0, h1, act, k = traj(bdt, bsteps)
    dh = h1[:, 0] - h0[:, 0]
    print(f"burn {it} dt {bdt} t/step {(time.time()-t1)/bsteps*1e3:.1f} ms alive {act.mean():.2f} k {k:.2f} H0 {np.median(h0[:,0]):.0f} qq {np.median(h0[:,1]):.0f} dH {np.median(dh):.1f}", flush=True)
    ctx.switch_partition()
q, p, xo, part = ctx.get_state()
print("u median", np.median(q[:, :4], 0), "sigma eps gamma", np.exp(np.median(q[:, :3], 0)))
for dt in (0.02, 0.05, 0.1, 0.15, 0.2, 0.3):
    ctx.set_state(q, None, xo, part)
    h0, h1, act, k = traj(dt, 8)
    dh = (h1[:, 0] - h0[:, 0])[act == 1]
    print(f"dt {dt}: alive {act.mean():.2f} k {k:.2f} dH med {np.median(dh):.3f} mean|dH| {np.mean(np.abs(dh)):.3f} acc {np.mean(np.minimum(1, np.exp(np.clip(-dh, -50, 0)))):.2f}", flush=True)
