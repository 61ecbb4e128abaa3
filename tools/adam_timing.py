"""Adam-based initial states for the boarding-school SIR configuration (BASELINE.json configs[3]): device-resident loop
against the host loop.  usage: python tools/adam_timing.py [chains] [host: 0/1] [sigma: 1.0 | variable]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.context import ChmcContext
from manifold_mcmc_for_diffusions_amd.workload import BOARDING_SCHOOL_COUNTS
from manifold_mcmc_for_diffusions_amd import init

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
host = len(sys.argv) > 2 and sys.argv[2] == "1"
sigma = sys.argv[3] if len(sys.argv) > 3 else "1.0"
sigma = "variable" if sigma == "variable" else float(sigma)
y = np.asarray(BOARDING_SCHOOL_COUNTS, dtype=np.float64)
ctx = ChmcContext("sir", 1.0, 200, len(y), y, sigma=sigma, num_chains=B)
print(f"sigma = {sigma}")
calls = {"n": 0, "t_obj": 0.0, "t_upd": 0.0}
_obj, _upd = ctx.adam_objective_device, ctx.adam_update_device
def obj(*a):
    t = time.perf_counter(); r = _obj(*a); calls["t_obj"] += time.perf_counter() - t; calls["n"] += 1
    return r
def upd(*a):
    t = time.perf_counter(); r = _upd(*a); calls["t_upd"] += time.perf_counter() - t
    return r
ctx.adam_objective_device, ctx.adam_update_device = obj, upd
for label, dr in (("device-resident", True),) + ((("host loop", False),) if host else ()):
    t0 = time.perf_counter()
    q, xo, tries = init.find_initial_states_by_gradient_descent_noisy_system(
        ctx, np.random.default_rng(20200710), adam_step_size=0.1, max_iters=5000, device_resident=dr)
    el = time.perf_counter() - t0
    print(f"{label}: {B} chains in {el:.2f} s; tries max {tries.max()} mean {tries.mean():.2f}; |c|max {np.abs(ctx.constr()).max():.1e}; "
          f"mean r^2 max {np.mean(q[:, -len(y):] ** 2, 1).max():.3f}")
    if calls["n"]:
        print(f"  {calls['n']} Adam iterations: objective + gradient {1e3 * calls['t_obj'] / calls['n']:.3f} ms, update "
              f"{1e3 * calls['t_upd'] / calls['n']:.3f} ms, everything else {1e3 * (el - calls['t_obj'] - calls['t_upd']) / calls['n']:.3f} ms "
              f"per iteration")
    calls.update(n=0, t_obj=0.0, t_upd=0.0)
