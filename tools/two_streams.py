"""Throughput of G independent half-batches driven from G host threads (one HIP stream each) against one batch.
usage: python tools/two_streams.py [total_chains] [groups] [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 32


def prepare(wl):
    for _ in range(5):
        wl.refresh_momentum()
        act = np.ones(wl.B, dtype=np.int32)
        for _ in range(16):
            r = wl.step(0.1, active=act)
            act &= (r["status"] == 0).astype(np.int32)
        wl.ctx.switch_partition()
    wl.refresh_momentum()


def run(wl, n):
    for k in range(n):
        if k and k % 16 == 0:
            wl.ctx.switch_partition()
            wl.refresh_momentum()
        wl.step(0.1)


wls = [None] * G
def make(i):
    wls[i] = FhnWorkload(num_chains=B // G, num_steps_per_obs=400, device=0, chain_offset=i * (B // G), total_chains=B)
    prepare(wls[i])
ths = [threading.Thread(target=make, args=(i,)) for i in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
ths = [threading.Thread(target=run, args=(wls[i], 4)) for i in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(wls[i], N)) for i in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
dt = time.perf_counter() - t0
print(f"{G} group(s) x {B // G} chains, {N} steps: {B * N / dt:.0f} steps/s ({dt / N * 1e3:.2f} ms per batch step)")
