"""Per-chain Newton iteration / status distribution on the boarding-school SIR bench workload (BASELINE configs[3]):
how many rounds of a lock-step batch serve how many chains.  usage: python tools/sir_iter_hist.py [chains] [step_size]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.workload import SirWorkload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
wl = SirWorkload(B, device=0)
ctx = wl.ctx
for _ in range(5):
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    for _ in range(16):
        r = wl.step(h, active=act)
        act &= (r["status"] == 0).astype(np.int32)
for masked in (False, True):
    print("=== trajectories", "ending at the first failed step" if masked else "with every chain stepping 16 times")
    for t in range(3):
        wl.refresh_momentum()
        act = np.ones(B, dtype=np.int32)
        hf, hb = np.zeros(52, int), np.zeros(52, int)
        st = np.zeros(5, int)
        mf, mb, na = [], [], []
        for k in range(16):
            r = wl.step(h, active=act if masked else None)
            on = act.astype(bool) if masked else np.ones(B, bool)
            f, b, s = r["iters_fwd"][on], r["iters_bwd"][on], r["status"][on]
            hf += np.bincount(np.minimum(f, 51), minlength=52)
            hb += np.bincount(np.minimum(b[s != 1], 51), minlength=52)
            st += np.bincount(s, minlength=5)[:5]
            mf.append(int(f.max()) if f.size else 0), mb.append(int(b.max()) if b.size else 0), na.append(int(on.sum()))
            if masked:
                act &= (r["status"] == 0).astype(np.int32)
        print(" traj", t, "status counts [ok, notconv, diverged, nonrev, -]:", st.tolist())
        print("  fwd iteration histogram:", {i: int(v) for i, v in enumerate(hf) if v})
        print("  bwd iteration histogram:", {i: int(v) for i, v in enumerate(hb) if v})
        print("  batch max fwd per step:", mf)
        print("  batch max bwd per step:", mb)
        print("  active chains per step:", na)
