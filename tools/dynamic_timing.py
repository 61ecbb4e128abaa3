"""Where the time of the batched dynamic (no-U-turn) transition goes at the bench operating point: wraps the context
calls and the torch bookkeeping of dynamic.DynamicTransition with synchronised timers.
usage: python tools/dynamic_timing.py [chains] [S] [transitions]"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload  # noqa: E402
from manifold_mcmc_for_diffusions_amd.dynamic import DynamicTransition  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n_tr = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wl = FhnWorkload(B, num_steps_per_obs=S, device_init=True)
ctx = wl.ctx
for it in range(5):  # burn-in towards the typical set, as bench.py does
    ctx.sample_momentum(wl.seed, 1000 + it)
    for _ in range(16):
        ctx.leapfrog_step(np.full(B, 0.1))
    ctx.switch_partition()
acc = defaultdict(float)
cnt = defaultdict(int)


def timed(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        cnt[name] += 1
        return r
    setattr(obj, name, g)


for nm in ("tree_begin", "tree_doubling_begin", "tree_step", "tree_doubling_end", "tree_get", "tree_get_doubling",
           "get_state_device", "restore_device", "sample_momentum", "switch_partition"):
    timed(ctx, nm)
tr = DynamicTransition(ctx, 0.09, seed=3, max_tree_depth=5)
import ctypes as C  # noqa: E402
from manifold_mcmc_for_diffusions_amd import _lib  # noqa: E402
L = _lib.lib()
if os.environ.get("CHMC_TIMING_CLASSES"):
    L.chmc_profile_enable(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 0
for it in range(n_tr):
    ctx.sample_momentum(3, it + 1)
    st = tr.sample(it)
    steps += int(st["n_step"].max())
    ctx.switch_partition()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{n_tr} transitions, {steps} batched leaves in {el:.2f} s = {el / steps * 1e3:.2f} ms per leaf; "
      f"{B * steps / el:.0f} leapfrog steps/s (upper bound: every chain live at every leaf)")
tot = 0.0
for k in acc:
    print(f"  {k:18s} {acc[k] / steps * 1e3:7.3f} ms per leaf  ({cnt[k]} calls, {acc[k] / cnt[k] * 1e3:.3f} ms each)")
    tot += acc[k]
print(f"  {'torch / numpy rest':18s} {(el - tot) / steps * 1e3:7.3f} ms per leaf")

if os.environ.get("CHMC_TIMING_CLASSES"):
    ms = np.zeros(10)
    nl = np.zeros(10, dtype=np.int64)
    L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), nl.ctypes.data_as(C.POINTER(C.c_longlong)))
    L.chmc_profile_enable(0)
    print("kernel classes, ms per leaf (launches per leaf): " + ", ".join(
        f"{k} {ms[i] / steps:.3f} ({nl[i] / steps:.1f})" for i, k in enumerate(_lib.KERNEL_CLASSES) if nl[i]))
    print(f"sum of classes {ms.sum() / steps:.3f} ms per leaf")
