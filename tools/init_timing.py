"""Host (NumPy) against device initial-state generation for the bench workload.  usage: python tools/init_timing.py [chains]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd import example_models as em
from manifold_mcmc_for_diffusions_amd.context import ChmcContext
from manifold_mcmc_for_diffusions_amd import init

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
y = em.simulate_fhn_observations(100, 0.2, 10000, seed=20200710, sigma=0.1)
ctx = ChmcContext("fhn", 0.2, 400, 5, y[:, 0], sigma=0.1, num_chains=B)
t0 = time.perf_counter()
q, xo, _ = init.fhn_initial_states(em.fhn, 0.2, 400, y, B, True)
t1 = time.perf_counter()
ctx.set_state(q, None, xo, 0)
t2 = time.perf_counter()
init.fhn_initial_states_device(ctx, em.fhn, y)
t3 = time.perf_counter()
qd = ctx.get_state()[0]
print(f"{B} chains: host solve {t1 - t0:.2f} s + upload/evaluate {t2 - t1:.3f} s; device (draws + solve + evaluate) "
      f"{t3 - t2:.3f} s; max |q_dev - q_host| = {np.abs(qd - q).max():.2e}")
