"""Leapfrog steps per second for the SIR model at BASELINE.json config 4's shape (T = 14 observations, S = 200 steps,
one block of R = 14 -> 16-row kernels), synthetic on-manifold states.  usage: python tools/sir_timing.py [chains] [S] [R]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd import example_models as em
from manifold_mcmc_for_diffusions_amd.context import ChmcContext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T, sigma, dt_obs = 14, 1.0, 0.25
R = int(sys.argv[3]) if len(sys.argv) > 3 else 14
m = em.MODELS["sir"]
rng = np.random.default_rng(3)
Q = m.dim_z + m.dim_v_0 + T * S * m.dim_v + T
q0 = np.zeros(Q)
q0[:4] = np.array([-1.0, -1.0, 1.0, 0.0]) + 0.1 * rng.standard_normal(4)
q0[4] = 1.0 + 0.1 * rng.standard_normal()
q0[5:5 + T * S * 3] = 0.3 * rng.standard_normal(T * S * 3)
q0[-T:] = rng.standard_normal(T)
# data generated from the state itself, so that the state lies on the constraint manifold
tmp = ChmcContext("sir", dt_obs, S, R, np.zeros(T), sigma=sigma, num_chains=1)
tmp.set_state(q0[None], None, np.zeros((1, T, 3)), 0)
tmp.update_x_obs_seq()
xo = tmp.get_state()[2][0]
tmp.close()
y = m.obs_func(xo)[:, 0] + sigma * q0[-T:]
ctx = ChmcContext("sir", dt_obs, S, R, y, sigma=sigma, num_chains=B)
ctx.set_state(np.repeat(q0[None], B, 0), None, np.repeat(xo[None], B, 0), 0)
print(f"SIR: {B} chains, dim_q = {ctx.Q}, blocks {ctx.num_blocks}, rows per block slot {ctx.RM}, |c|max = {np.abs(ctx.constr()).max():.1e}")
ctx.sample_momentum(1, 1)
dt = np.full(B, 0.02)
for _ in range(3):
    r = ctx.leapfrog_step(dt)
n = 20
from manifold_mcmc_for_diffusions_amd import _lib
import ctypes as C
L = _lib.lib()
L.chmc_profile_enable(1)
t0 = time.perf_counter()
for _ in range(n):
    r = ctx.leapfrog_step(dt)
el = time.perf_counter() - t0
print(f"{n} steps: {el / n * 1e3:.2f} ms per batched step = {B * n / el:.0f} steps/s; ok {np.mean(r['status'] == 0):.3f}, "
      f"Newton iterations fwd {r['iters_fwd'].mean():.2f} bwd {r['iters_bwd'].mean():.2f}")
ms = np.zeros(10)
nl = np.zeros(10, dtype=np.int64)
L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), nl.ctypes.data_as(C.POINTER(C.c_longlong)))
L.chmc_profile_enable(0)
print("kernel classes, ms per step (launches per step): " + ", ".join(
    f"{k} {ms[i] / n:.2f} ({nl[i] / n:.0f})" for i, k in enumerate(_lib.KERNEL_CLASSES) if nl[i]))
