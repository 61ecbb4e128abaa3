"""Worker of tests/test_distributed_gloo.py: one rank of a chain-sharded run on CPU (gloo), driving the library's
host logic through the TEST-ONLY emulation build.  argv: out_file total_chains n_steps"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd import _lib, distributed as D, example_models as em  # noqa: E402
from manifold_mcmc_for_diffusions_amd.context import ChmcContext  # noqa: E402
from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states  # noqa: E402


def run(total, n_steps, rank, world):
    _lib._LIB = _lib._bind(ctypes.CDLL(os.path.join(HERE, "emu", "libchmc_emu.so")))
    T, S, R, sigma = 8, 4, 3, 0.1
    y = em.simulate_fhn_observations(T, 0.2, 50, seed=11, sigma=sigma)
    off, cnt = D.shard_chains(total, rank, world)
    q, xo, rngs = fhn_initial_states(em.fhn, 0.2, S, y, cnt, True, seed=13, chain_offset=off, total_chains=total)
    ctx = ChmcContext("fhn", 0.2, S, R, y[:, 0], sigma=sigma, num_chains=cnt)
    ctx.set_state(q, np.stack([r.standard_normal(ctx.Q) for r in rngs]), xo, 0)
    ctx.project_onto_cotangent_space()
    dt = np.where((np.arange(cnt) + off) % 2 == 0, 0.03, -0.03)
    for _ in range(n_steps):
        ctx.leapfrog_step(dt, constraint_tol=1e-9, position_tol=1e-8)
    ctx.switch_partition()
    qf, pf, _, _ = ctx.get_state()
    ham = ctx.hamiltonian()
    local = np.concatenate([qf[:, :6], pf[:, :3], ham[:, :1]], 1)
    ctx.close()
    return local


def run_sampler(total, n_iter, rank, world, metric=False):
    """Static-trajectory sampler with step-size adaptation on this rank's shard (cross-rank accept statistic);
    metric: also the block-diagonal metric adapter (per-chain statistics combined over the chains of all ranks)."""
    from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc
    from manifold_mcmc_for_diffusions_amd.adapters import OnlineBlockDiagonalMetricAdapter
    _lib._LIB = _lib._bind(ctypes.CDLL(os.path.join(HERE, "emu", "libchmc_emu.so")))
    T, S, R, sigma = 6, 4, 2, 0.1
    y = em.simulate_fhn_observations(T, 0.2, 50, seed=5, sigma=sigma)
    off, cnt = D.shard_chains(total, rank, world)
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, S, y, cnt, True, seed=7, chain_offset=off, total_chains=total)
    ctx = ChmcContext("fhn", 0.2, S, R, y[:, 0], sigma=sigma, num_chains=cnt)
    ctx.set_state(q, None, xo, 0)
    res = sample_static_chmc(ctx, n_iter, 2, 0.05, seed=3, chain_offset=off, n_adapt=n_iter - 2, total_chains=total,
                             metric_adapter=OnlineBlockDiagonalMetricAdapter(4) if metric else None)
    ctx.close()
    hist = np.concatenate([res["step_size"], res["accept_stat"], [res["final_step_size"]]])
    if metric:
        hist = np.concatenate([hist, res["metric_M_0"].ravel()])
    return np.concatenate([res["heads"][-1], np.tile(hist, (cnt, 1))], 1)


if __name__ == "__main__":
    out, total, n_steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, _, world = D.init_process_group("gloo")
    if len(sys.argv) > 4 and sys.argv[4] in ("sampler", "sampler_metric"):
        local = run_sampler(total, n_steps, rank, world, metric=sys.argv[4] == "sampler_metric")
    else:
        local = run(total, n_steps, rank, world)
    gathered = D.gather_samples(local)
    tmax = D.max_over_ranks(float(rank + 1))
    tot = D.sum_over_ranks([local.shape[0]])
    if rank == 0:
        assert gathered.shape[0] == total and tmax == float(world) and int(tot[0]) == total
        np.save(out, gathered)
    D.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
