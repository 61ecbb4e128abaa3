"""One leapfrog step from identical states with the sequential and with the time-parallel forward scan: per-chain status,
iteration counts and results side by side.  usage: python tools/par_scan_compare.py [chains] [S] [step size] [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
from manifold_mcmc_for_diffusions_amd.context import ChmcContext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 10
os.environ["CHMC_PAR_SCAN"] = "0"
wl = SirWorkload(B, num_steps_per_obs=S)
a = wl.ctx
os.environ["CHMC_PAR_SCAN"] = "1"
b = ChmcContext("sir", wl.obs_interval, S, wl.R, wl.y[:, 0], sigma=1.0, num_chains=B)
for _ in range(4):  # burn-in on the sequential context: the Adam initial states are atypically benign
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    for _ in range(12):
        r0 = a.leapfrog_step(dt, active=act)
        act &= (r0["status"] == 0).astype(np.int32)
tot = dict(n=0, status_diff=0, ok_a=0, ok_b=0, iter_diff=0, maxdq=0.0)
for r in range(rounds):
    wl.refresh_momentum()
    q, p, xo, part = a.get_state()
    b.set_state(q, p, xo, part)
    b.project_onto_cotangent_space()  # (p is already tangent; marks it so)
    a.set_state(q, p, xo, part)
    a.project_onto_cotangent_space()
    ra, rb = a.leapfrog_step(dt), b.leapfrog_step(dt)
    qa, qb = a.get_state()[0], b.get_state()[0]
    both = (ra["status"] == 0) & (rb["status"] == 0)
    tot["n"] += B
    tot["status_diff"] += int((ra["status"] != rb["status"]).sum())
    tot["ok_a"] += int((ra["status"] == 0).sum())
    tot["ok_b"] += int((rb["status"] == 0).sum())
    tot["iter_diff"] += int(((ra["iters_fwd"] != rb["iters_fwd"]) | (ra["iters_bwd"] != rb["iters_bwd"]))[both].sum())
    if both.any():
        tot["maxdq"] = max(tot["maxdq"], float(np.abs(qa[both] - qb[both]).max() / max(1.0, np.abs(qa[both]).max())))
    d = np.flatnonzero(ra["status"] != rb["status"])
    if d.size:
        print(f"round {r}: status differs for chains {d.tolist()}: sequential {ra['status'][d].tolist()} iters {ra['iters_fwd'][d].tolist()}/{ra['iters_bwd'][d].tolist()}, "
              f"time-parallel {rb['status'][d].tolist()} iters {rb['iters_fwd'][d].tolist()}/{rb['iters_bwd'][d].tolist()}")
    for _ in range(3):  # move on (sequential context) before the next comparison
        a.leapfrog_step(dt)
import ctypes as C
from manifold_mcmc_for_diffusions_amd import _lib
L = _lib.lib()
cnt = b.diagnostics()["par_scan"]
print(tot, "sequential fallbacks", cnt[0])
