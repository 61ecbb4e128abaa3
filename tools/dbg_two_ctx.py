"""Experiment: do two half-size contexts driven from two host threads (two HIP streams) overlap usefully?"""
import sys, os, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 256 // NT
K = 16
dev = torch.device("cuda:0")
bar = threading.Barrier(NT + 1)
res = {}
def worker(i):
    wl = FhnWorkload(B, num_steps_per_obs=400, chain_offset=i * B, total_chains=256)
    for _ in range(4):
        p = np.stack([r.standard_normal(wl.ctx.Q) for r in wl.rngs]); wl.ctx.set_momentum(p); wl.ctx.project_onto_cotangent_space()
        for _ in range(16): wl.step(0.1)
        wl.ctx.switch_partition()
    p = np.stack([r.standard_normal(wl.ctx.Q) for r in wl.rngs]); wl.ctx.set_momentum(p); wl.ctx.project_onto_cotangent_space()
    bar.wait()
    t0 = time.perf_counter()
    for _ in range(K): wl.step(0.1)
    res[i] = time.perf_counter() - t0
    bar.wait()
ths = [threading.Thread(target=worker, args=(i,)) for i in range(NT)]
for t in ths: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); el = time.perf_counter() - t0
for t in ths: t.join()
print(f"threads {NT} x {B} chains: {256*K/el:.0f} steps/s  ({el/K*1e3:.2f} ms per 256-chain step); per-thread {res}")
