import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ctypes as C
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
from manifold_mcmc_for_diffusions_amd import _lib
wl = FhnWorkload(256, num_steps_per_obs=400)
ctx = wl.ctx; L = _lib.lib()
def prof():
    ms = np.zeros(10); n = np.zeros(10, dtype=np.int64)
    L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), n.ctypes.data_as(C.POINTER(C.c_longlong)))
    return {k: (round(ms[i] / max(n[i], 1), 4), int(n[i])) for i, k in enumerate(_lib.KERNEL_CLASSES) if n[i]}
L.chmc_profile_enable(1)
for _ in range(5): ctx.constr()
print("constr only (no traj store):", prof())
L.chmc_profile_enable(1)
t0 = time.time()
for _ in range(3): ctx.update_x_obs_seq()
print("xobs wall", (time.time() - t0) / 3)
