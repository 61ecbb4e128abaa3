"""Time per batched leapfrog step in each partition of the bench workload.  usage: python tools/partition_timing.py [chains]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = FhnWorkload(num_chains=B, num_steps_per_obs=400, device=0, device_init=True)
ctx = wl.ctx
for _ in range(5):
    wl.refresh_momentum()
    act = np.ones(B, dtype=np.int32)
    for _ in range(16):
        r = wl.step(0.1, active=act)
        act &= (r["status"] == 0).astype(np.int32)
    ctx.switch_partition()
for rep in range(4):
    wl.refresh_momentum()
    t0 = time.perf_counter()
    for _ in range(16):
        r = wl.step(0.1)
    dt = time.perf_counter() - t0
    print(f"partition {ctx.partition} (K = {ctx.num_blocks}): {dt / 16 * 1e3:.2f} ms per step, iterations fwd max "
          f"{r['iters_fwd'].max()} bwd max {r['iters_bwd'].max()}")
    ctx.switch_partition()
