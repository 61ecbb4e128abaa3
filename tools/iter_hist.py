"""Distribution of Newton iterations per chain and per batch step on the bench workload (how much of a step is spent
on the last few chains of the batch).  usage: python tools/iter_hist.py [chains]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = FhnWorkload(num_chains=B, num_steps_per_obs=400, device=0)
ctx = wl.ctx
for _ in range(5):
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    for _ in range(16):
        r = wl.step(0.1, active=act)
        act &= (r["status"] == 0).astype(np.int32)
    ctx.switch_partition()
wl.refresh_momentum()
hf, hb = np.zeros(12, int), np.zeros(12, int)
mf, mb = [], []
for k in range(16):
    r = wl.step(0.1)
    f, b = r["iters_fwd"], r["iters_bwd"]
    hf += np.bincount(np.minimum(f, 11), minlength=12)
    hb += np.bincount(np.minimum(b, 11), minlength=12)
    mf.append(int(f.max())), mb.append(int(b.max()))
print("chains x steps by forward iterations :", hf.tolist())
print("chains x steps by backward iterations:", hb.tolist())
print("batch max per step fwd:", mf)
print("batch max per step bwd:", mb)
print("mean per chain fwd %.2f bwd %.2f; mean batch max fwd %.2f bwd %.2f" % (
    (hf * np.arange(12)).sum() / hf.sum(), (hb * np.arange(12)).sum() / hb.sum(), np.mean(mf), np.mean(mb)))
