"""A minimal batched sampler on top of the device step: static-length (optionally jittered) constrained HMC
trajectories with a Metropolis accept step, momentum refresh, partition switching and dual-averaging step-size
adaptation.

The reference drives the same integrator with Mici's dynamic multinomial (NUTS-style) transition
(scripts/utils.py:292-306); the batched counterpart of that transition is dynamic.py.  This static variant targets the
same posterior (it is a valid Markov kernel for it), costs no per-leaf bookkeeping and is what bench-like workloads
use: momentum -> L constrained leapfrog steps -> accept / reject -> SwitchPartitionTransition."""
import os
import numpy as np


class DualAveragingStepSize:
    """Hoffman & Gelman dual averaging on the mean accept statistic (cf. mici DualAveragingStepSizeAdapter,
    scripts/utils.py:303-306), one step size shared by all chains."""

    def __init__(self, step_size, target=0.8, gamma=0.05, t0=10.0, kappa=0.75):
        self.mu = np.log(10 * step_size)
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.h_bar, self.log_bar, self.t = 0.0, np.log(step_size), 0
        self.step_size = step_size

    def update(self, accept_stat):
        self.t += 1
        eta = 1.0 / (self.t + self.t0)
        self.h_bar = (1 - eta) * self.h_bar + eta * (self.target - accept_stat)
        log_eps = self.mu - np.sqrt(self.t) / self.gamma * self.h_bar
        w = self.t ** (-self.kappa)
        self.log_bar = w * log_eps + (1 - w) * self.log_bar
        self.step_size = float(np.exp(log_eps))
        return self.step_size

    def final(self):
        return float(np.exp(self.log_bar))


class PerChainDualAveragingStepSize:
    """The same adapter with one state per chain, as Mici runs it: every chain adapts ITS OWN step size on its own accept
    statistic during warm-up (mici DualAveragingStepSizeAdapter keeps one adapter state per chain), and the step size of
    the main phase is the average of the chains' smoothed step sizes (Mici's `finalize` with several chains, as recalled:
    SURVEY.md Appendix D marks the adapter defaults `medium`).  A chain that starts where the shared step size is far too
    long for its retraction (every first step an integrator error, accept statistic 0) thereby shortens its own steps
    until it moves, instead of staying put for the whole run while dragging a shared step size down."""

    def __init__(self, step_size, num_chains, target=0.8, gamma=0.05, t0=10.0, kappa=0.75):
        self.mu = np.full(num_chains, np.log(10 * step_size))
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.h_bar = np.zeros(num_chains)
        self.log_bar = np.full(num_chains, np.log(step_size))
        self.t = 0
        self.step_size = np.full(num_chains, float(step_size))

    def update(self, accept_stat):
        self.t += 1
        eta = 1.0 / (self.t + self.t0)
        self.h_bar = (1 - eta) * self.h_bar + eta * (self.target - np.asarray(accept_stat, dtype=np.float64))
        log_eps = self.mu - np.sqrt(self.t) / self.gamma * self.h_bar
        w = self.t ** (-self.kappa)
        self.log_bar = w * log_eps + (1 - w) * self.log_bar
        self.step_size = np.exp(log_eps)
        return self.step_size

    def final(self):
        """Smoothed step size of every chain [B]; the caller averages over the chains of all ranks."""
        return np.exp(self.log_bar)


def _mean_over_all_chains(local_sum, local_count):
    """Mean of a per-chain statistic over the chains of every rank (one tiny all-reduce; SURVEY.md 8f #3: warm-up
    adaptation combines the chains of all GPUs so that every rank integrates with the same step size)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import distributed as D
            tot = D.sum_over_ranks([float(local_sum), float(local_count)])
            return float(tot[0]) / float(tot[1])
    except ImportError:
        pass
    return float(local_sum) / float(local_count)


def sample_static_chmc(ctx, n_iter, n_step, step_size, seed, chain_offset=0, n_adapt=0, solver=None, rng=None,
                       n_head=6, callback=None, trace_dir=None, trace_func=None, total_chains=None,
                       jitter_length=False, metric_adapter=None, metric_window=0.75, metric_skip=0.25):
    """Runs n_iter transitions on all chains of `ctx`; returns traces of the first `n_head` position components
    ([n_iter, B, n_head]), accept statistics and the step size used.  Directions are sampled per chain and
    transition (forward / backward in time), failed trajectories are rejected.

    trace_dir: write memory-mapped traces (`trace_<var>.npy`, [chain, draw, ...]) and `summary.json` there;
    trace_func(head [B, n_head], hamiltonian [B]) -> {name: [B, ...]} chooses the traced variables (default: the
    head components as `pos_head` and `hamiltonian`).  Under torch.distributed the accept statistic that drives the
    step-size adaptation is averaged over the chains of all ranks; with `total_chains` (the number of chains of the
    whole job) the per-chain directions and accept draws are taken from one stream indexed by the global chain
    number, so that any sharding of the chains reproduces the single-process run.
    jitter_length: every chain draws its number of leapfrog steps uniformly from 1..n_step per transition (a mixture
    of reversible kernels, so still a valid sampler); chains in regions where the retraction often fails then still
    move with their short trajectories instead of rejecting every long one (Mici's dynamic transition gets the same
    effect by ending a trajectory at the failing step).
    metric_adapter: an `adapters.OnlineBlockDiagonalMetricAdapter` over the `dim_u` global parameters (standard
    splitting only).  It sees the warm-up draws between the fractions `metric_skip` (the initial transient is left
    out, like the first window of a staged warm-up) and `metric_window` of the warm-up; the block metric it produces
    (per-chain statistics combined over all chains and ranks) is installed with `ctx.set_metric` there, and the
    remaining warm-up transitions re-tune the step size for the new metric."""
    solver = dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10, max_iters=50,
                  reverse_check_tol=2e-8) if solver is None else solver
    rng = np.random.default_rng(seed) if rng is None else rng
    adapter = DualAveragingStepSize(step_size) if n_adapt > 0 else None
    B = ctx.B

    def draw():  # uniforms of this rank's chains
        if total_chains is None:
            return rng.random(B)
        return rng.random(total_chains)[chain_offset:chain_offset + B]

    n_metric, n_skip, metric_state = 0, 0, None
    if metric_adapter is not None and n_adapt > 0:
        if metric_adapter.dim_param != ctx.U:
            raise ValueError(f"the block metric covers the dim_u = {ctx.U} global parameters")
        n_metric = max(2, int(metric_window * n_adapt))
        n_skip = min(int(metric_skip * n_adapt), n_metric - 2)
        n_head = max(n_head, ctx.U)
    stuck = np.zeros(B, dtype=np.int64)  # consecutive trajectories with zero acceptance probability
    # per chain: how its post-warm-up transitions ended: [accepted, rejected after a complete trajectory, retraction did
    # not converge (status 1), diverged (2), non-reversible step (3)] -- the diagnosis of a chain that does not move
    outcome = np.zeros((B, 5), dtype=np.int64)
    heads = np.empty((n_iter, B, n_head))
    acc_hist, eps_hist, fail_hist = np.empty(n_iter), np.empty(n_iter), np.empty(n_iter)
    if trace_func is None:
        def trace_func(head, ham):
            return {"pos_head": head, "hamiltonian": ham}
    writer, t_start, c_start = None, None, None
    if trace_dir is not None:
        import time
        from .traces import TraceWriter
        t_start, c_start = time.perf_counter(), ctx.counters()
    for it in range(n_iter):
        ctx.sample_momentum(seed, it + 1, chain_offset)
        h0 = ctx.hamiltonian()[:, 0]
        ctx.snapshot()
        # warm-up only: a chain whose trajectories keep failing (a start far from the typical set, where the shared
        # step size is too long for the retraction) backs its own step size off until it moves again
        scale = 0.5 ** np.minimum(stuck, 8) if it < n_adapt else 1.0
        dt = np.where(draw() < 0.5, step_size, -step_size) * scale
        act = np.ones(B, dtype=np.int32)
        # the trajectory as ONE library call (the loop an integration transition runs around integrator.step,
        # scripts/utils.py:284-301: a chain stops at its first failed step): single-block layouts walk it in one launch per
        # chain, the lock-step path folds the kicks of consecutive steps when every chain takes the same number of steps
        length = 1 + np.floor(draw() * n_step).astype(np.int32) if jitter_length else int(n_step)
        r = ctx.leapfrog_steps(dt, length, active=act, **solver)
        end_status = np.where(r["status"] > 0, r["status"], 0).astype(np.int64)
        act &= (r["status"] == 0).astype(np.int32)
        h1 = ctx.hamiltonian()[:, 0]
        dh = h1 - h0
        prob = np.where((act == 1) & np.isfinite(dh), np.exp(np.minimum(0.0, -np.where(np.isfinite(dh), dh, np.inf))), 0.0)
        accept = draw() < prob
        stuck = np.where(prob > 0.0, 0, stuck + 1)
        if it >= n_adapt:
            col = np.where(end_status > 0, 1 + np.clip(end_status, 1, 3), np.where(accept, 0, 1))
            outcome[np.arange(B), col] += 1
        ctx.restore((~accept).astype(np.int32))
        ctx.switch_partition()
        heads[it] = ctx.get_head(n_head)
        if trace_dir is not None:
            vals = {k: np.asarray(v) for k, v in trace_func(heads[it], ctx.hamiltonian()[:, 0]).items()}
            if writer is None:
                writer = TraceWriter(trace_dir, B, n_iter, {k: v.shape[1:] for k, v in vals.items()})
            writer.write(it, vals)
        if n_metric and n_skip <= it < n_metric:
            if metric_state is None:
                metric_state = metric_adapter.initialize(np.zeros((B, ctx.Q)))
            metric_adapter.update(metric_state, heads[it])
            if it == n_metric - 1:
                metric = metric_adapter.finalize(metric_state)
                ctx.set_metric(metric.blocks[0].array)
                adapter = DualAveragingStepSize(step_size)  # the step size is re-tuned under the new metric
        acc_all = _mean_over_all_chains(prob.sum(), B)
        acc_hist[it], eps_hist[it], fail_hist[it] = acc_all, step_size, 1.0 - act.mean()
        if adapter is not None and it < n_adapt:
            step_size = adapter.update(acc_all)
            if it == n_adapt - 1:
                step_size = adapter.final()
        if callback is not None:
            callback(it, heads[it], prob.mean(), step_size)
    out = dict(heads=heads, accept_stat=acc_hist, step_size=eps_hist, fail_rate=fail_hist, final_step_size=step_size,
               chain_outcomes=outcome)
    if n_metric:
        out["metric_M_0"] = None if ctx.M_0 is None else ctx.M_0.copy()
    if writer is not None:
        import time
        from .traces import save_summary
        writer.flush()
        c_end = ctx.counters()
        main = {k: np.asarray(v)[:, n_adapt:] for k, v in writer.arrays().items()}  # summary over the post-warm-up draws
        out["summary"] = save_summary(trace_dir, main, None, time.perf_counter() - t_start, step_size,
                                      {k: c_end[k] - c_start[k] for k in c_end if k != "_"})
        out["trace_files"] = {k: os.path.join(trace_dir, f"trace_{k}.npy") for k in writer.arrays()}
    return out
