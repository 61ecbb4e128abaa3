"""Time of one value + gradient evaluation of the unconstrained-HMC target (chmc_neg_log_dens_and_grad) at the bench
workload's size.  usage: python tools/hmc_target_timing.py [chains] [S]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd import example_models as em
from manifold_mcmc_for_diffusions_amd.context import ChmcContext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
y = em.simulate_fhn_observations(100, 0.2, 10000, seed=20200710, sigma=0.1)
ctx = ChmcContext("fhn", 0.2, S, 5, y[:, 0], sigma=0.1, num_chains=B)
rng = np.random.default_rng(1)
q = 0.1 * rng.standard_normal((B, ctx.U + ctx.NV))
q[:, :4] += np.array([-1.2, -2.0, 0.4, 0.8])
from manifold_mcmc_for_diffusions_amd import _lib
import ctypes as C
L = _lib.lib()
ctx.neg_log_dens_and_grad(q)
L.chmc_profile_enable(1)
t0 = time.perf_counter()
for _ in range(5):
    val, g = ctx.neg_log_dens_and_grad(q)
el = (time.perf_counter() - t0) / 5
ms = np.zeros(10); nl = np.zeros(10, dtype=np.int64)
L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), nl.ctypes.data_as(C.POINTER(C.c_longlong)))
L.chmc_profile_enable(0)
print(f"{B} chains, S = {S}: {el * 1e3:.1f} ms per call including the PCIe round trip of q and grad ({2 * q.nbytes / 1e6:.0f} MB); "
      f"device: forward scan {ms[7] / 5:.2f} ms, adjoint sweep {ms[0] / 5:.2f} ms; finite {np.isfinite(val).all()}")
