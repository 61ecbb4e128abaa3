"""Batched dynamic-integration-time transition: the caller of the hot path in the reference's experiments
(mici.transitions.MultinomialDynamicIntegrationTransition, wired in scripts/utils.py:292-301).

Mici builds the trajectory tree of one chain recursively.  Here every chain of a context grows its own tree in lock
step: at tree depth d every live chain picks its own direction, resumes from its own tree edge and takes 2^d leapfrog
steps; multinomial sampling of the proposal (uniform within a sub-tree, biased progressive between the old tree and a
new sub-tree), the no-U-turn criterion `dh_dmom(edge) . sum_mom < 0` on every sub-tree span (checked iteratively with
O(depth) momentum checkpoints per chain instead of recursion), termination on integrator errors and on divergence
(`delta_h > max_delta_h`) follow the reference's transition, and so do Mici's additional sub-tree checks
(`do_extra_subtree_checks`, on by default there): for every sub-tree span of four or more leaves the criterion is also
applied from the span's first leaf to the first leaf of its right half (left half's momenta plus that leaf's) and from
the last leaf of its left half to the span's last leaf (right half's momenta plus that leaf's).
Tree vectors (edges, proposal, momentum sums, checkpoints) live in torch tensors on the context's device -- plumbing
around the integrator step, which does all the work -- and the library re-evaluates a chain's state caches when its
tree switches edges, one batched evaluation per doubling at most.  Every per-chain decision is taken on the device: those
of a leaf in `chmc_tree_step` (one library call and one 4-byte read-back per leaf), those of a doubling -- direction, edge
switch, biased progressive sampling, whole-tree criterion -- in `chmc_tree_doubling_begin / _end` (two calls and 16 bytes
per doubling); the host draws the keyed uniforms and loops.

All random choices come from `TreeUniforms`, keyed by (seed, transition, purpose, depth, leaf) and the global chain
index, so they do not depend on how chains are sharded or on the order in which an implementation asks for them.  The
chains' arithmetic is sharding-independent bit for bit as well: the library chooses its kernels (time-parallel or
sequential forward scan, interval-parallel or stored-rows 16-row sweeps, the per-chain retraction kernel) from the layout
alone -- blocks per chain, block length, rows -- never from the number of chains in the context (csrc/chmc_api.inc
few_long_blocks; GPU test test_results_do_not_depend_on_the_shard_size).
"""
import numpy as np


class TreeUniforms:
    """U(0, 1) draws of one transition: `get(kind, depth, leaf)` -> [B] for this rank's chains."""
    DIRECTION, ACCEPT, LEAF = 0, 1, 2

    def __init__(self, seed, transition, total_chains, chain_offset, num_chains):
        self.key = (int(seed), int(transition))
        self.total, self.off, self.B = total_chains, chain_offset, num_chains

    def get(self, kind, depth, leaf=0):
        rng = np.random.default_rng(self.key + (int(kind), int(depth), int(leaf)))
        return rng.random(self.total)[self.off:self.off + self.B]


def _ckpt_range(k):
    """Checkpoint slots touched by leaf k of a sub-tree (Phan & Pradhan's iterative no-U-turn check): idx_max = number
    of set bits of k >> 1, idx_min = idx_max - (number of trailing set bits of k) + 1."""
    idx_max = bin(k >> 1).count("1")
    trailing = 0
    n = k
    while n & 1:
        n >>= 1
        trailing += 1
    return idx_max - trailing + 1, idx_max


class DynamicTransition:
    """state = the context's current chain states; `sample(it)` runs one transition for every chain and leaves the
    selected states set on the context (momentum to be refreshed by the caller)."""

    def __init__(self, ctx, step_size, seed, max_tree_depth=10, max_delta_h=1000.0, solver=None, chain_offset=0,
                 total_chains=None, device=None, do_extra_subtree_checks=True):
        import torch
        self.torch = torch
        self.ctx, self.seed = ctx, seed
        self.step_size = step_size  # a number, or one step size per chain [B] (per-chain warm-up adaptation)
        self.max_tree_depth, self.max_delta_h = int(max_tree_depth), float(max_delta_h)
        self.solver = dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10, max_iters=50,
                           reverse_check_tol=2e-8) if solver is None else solver
        self.off = chain_offset
        self.total = ctx.B + chain_offset if total_chains is None else total_chains
        if device is None:
            device = torch.device("cuda", 0) if ctx.L.chmc_backend().startswith(b"hip") else torch.device("cpu")
        self.dev = device
        B, Q = ctx.B, ctx.Q
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=device)  # noqa: E731
        self.q, self.p = z(B, Q), z(B, Q)                       # staging of the context's current state
        self.neg_q, self.neg_p, self.pos_q, self.pos_p = z(B, Q), z(B, Q), z(B, Q), z(B, Q)
        self.prop_q, self.sub_prop_q = z(B, Q), z(B, Q)
        self.sum_mom, self.sub_sum = z(B, Q), z(B, Q)
        self.ck_p = z(self.max_tree_depth, B, Q)                # momentum at the first leaf of a pending span
        self.ck_sum = z(self.max_tree_depth, B, Q)              # running sub-tree momentum sum at that leaf
        # momentum at the last leaf of the left half of a pending span (additional sub-tree checks)
        self.ck_end = z(self.max_tree_depth, B, Q) if do_extra_subtree_checks else None

    # ---- helpers
    def _sync(self):
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize(self.dev)

    def _fetch(self):
        self._sync()
        self.ctx.get_state_device(self.q.data_ptr(), self.p.data_ptr())

    def _mask(self, m):
        return self.torch.from_numpy(np.ascontiguousarray(m)).to(self.dev)

    def _dot(self, edge, span):
        """dh_dmom(edge) . span per chain: dh_dmom = metric.inv @ mom (sde/mici_extensions.py:1204-1208); with the block
        metric only the u-part differs from the plain inner product."""
        d = (edge * span).sum(1)
        M0 = getattr(self.ctx, "M_0", None)
        if M0 is not None:
            U = M0.shape[0]
            Wm = self.torch.from_numpy(np.linalg.inv(M0) - np.eye(U)).to(self.dev)
            d = d + ((edge[:, :U] @ Wm) * span[:, :U]).sum(1)
        return d

    def sample(self, it):
        torch, ctx = self.torch, self.ctx
        B = ctx.B
        un = TreeUniforms(self.seed, it, self.total, self.off, B)
        self._fetch()
        # Per-chain tree state lives in the library (chmc_tree_*): the decisions of a leaf (integrator error, divergence,
        # multinomial weight and proposal, no-U-turn termination of sub-tree spans: chmc_tree_step) and of a doubling
        # (direction, edge switch with cache re-evaluation, biased progressive sampling, whole-tree criterion:
        # chmc_tree_doubling_begin / _end) are taken on the device; the host draws the keyed uniforms, chooses the sign of
        # the step by the same comparison, and reads back 4-8 bytes per call.
        h0 = ctx.tree_begin()
        for t in (self.neg_q, self.pos_q, self.prop_q):
            t.copy_(self.q)
        for t in (self.neg_p, self.pos_p, self.sum_mom):
            t.copy_(self.p)
        self._sync()  # the library works on its own stream
        ck_end = None if self.ck_end is None else self.ck_end.data_ptr()
        edges = (self.neg_q.data_ptr(), self.neg_p.data_ptr(), self.pos_q.data_ptr(), self.pos_p.data_ptr())
        for d in range(self.max_tree_depth):
            u_dir = un.get(un.DIRECTION, d)
            if ctx.tree_doubling_begin(u_dir, *edges, self.sub_sum.data_ptr()) == 0:
                break
            dt = np.where(u_dir < 0.5, self.step_size, -np.asarray(self.step_size))
            # ---- sub-tree of 2^d leaves: one call per leaf (integrator step + momentum sum, multinomial proposal,
            # checkpoint of an even leaf, for an odd leaf the no-U-turn checks over every sub-tree span that ends there)
            for k in range(1 << d):
                lo, hi = _ckpt_range(k)
                even = k % 2 == 0
                n_run = ctx.tree_step(dt, un.get(un.LEAF, d, k), self.max_delta_h, self.sub_prop_q.data_ptr(),
                                      self.sub_sum.data_ptr(), self.ck_p.data_ptr(), self.ck_sum.data_ptr(), ck_end,
                                      hi if even else -1, lo, 0 if even else hi - lo + 1, **self.solver)
                if n_run == 0:
                    break
            if ctx.tree_doubling_end(un.get(un.ACCEPT, d), d, self.prop_q.data_ptr(), self.sub_prop_q.data_ptr(),
                                     self.sum_mom.data_ptr(), self.sub_sum.data_ptr(), *edges) == 0:
                break
        st = ctx.tree_get()
        dbl = ctx.tree_get_doubling()
        # leave the selected positions on the context (every chain: the context may sit on a tree edge)
        ctx.restore_device(self.prop_q.data_ptr(), self.p.data_ptr(), np.ones(B, dtype=np.int32), False)
        n_step = st["n_step"].astype(np.int64)
        return dict(accept_stat=st["sum_acc"] / np.maximum(n_step, 1), n_step=n_step, depth=dbl["depth"], moved=dbl["moved"],
                    diverged=st["diverged"] != 0, integrator_error=st["failed"] != 0, h0_finite=np.isfinite(h0))


def sample_dynamic_chmc(ctx, n_iter, step_size, seed, n_adapt=0, chain_offset=0, total_chains=None, n_head=6,
                        trace_dir=None, trace_func=None, callback=None, per_chain_step_size=True, **kw):
    """Momentum refresh -> dynamic transition -> partition switch, `n_iter` times for all chains of `ctx`, with
    dual-averaging step-size adaptation on the tree's accept statistic during warm-up: the reference's sampling loop
    (scripts/utils.py:292-306, 338-365), batched.  As in Mici every chain adapts its own step size during warm-up (the
    integrator takes a step size per chain); the main phase runs with the average of the chains' adapted step sizes,
    taken over the chains of all ranks.  per_chain_step_size=False: ONE step size adapted on the accept statistic
    averaged over all chains (round 2's scheme): the batched trees of a doubling then all have the same number of leaves to
    offer, which keeps the lock-step batch full during warm-up (the FitzHugh-Nagumo example: 30 k against 24 k leapfrog
    steps/s end to end), but a chain that starts where that step size is far too long never moves."""
    import time
    from .sampling import DualAveragingStepSize, PerChainDualAveragingStepSize, _mean_over_all_chains
    tr = DynamicTransition(ctx, step_size, seed, chain_offset=chain_offset, total_chains=total_chains, **kw)
    B = ctx.B
    adapter, adapted = None, None
    if n_adapt > 0:
        adapter = PerChainDualAveragingStepSize(step_size, B) if per_chain_step_size else DualAveragingStepSize(step_size)
    heads = np.empty((n_iter, B, n_head))
    acc_hist, eps_hist, nstep_hist = np.empty(n_iter), np.empty(n_iter), np.empty(n_iter)
    err_hist = np.empty(n_iter)
    # per chain, main phase (after warm-up): transitions that moved the chain / did not, trees ended by an integrator error,
    # by a divergence, trees with no leaf at all (first step failed), non-finite root Hamiltonian; leaves and accept-stat sums
    outcome = np.zeros((B, 6), dtype=np.int64)
    per_chain = dict(n_step=np.zeros(B), accept=np.zeros(B), err=np.zeros(B))
    writer, t0, c0 = None, time.perf_counter(), ctx.counters()
    if trace_func is None:
        def trace_func(head, ham):
            return {"pos_head": head, "hamiltonian": ham}
    for it in range(n_iter):
        ctx.sample_momentum(seed, it + 1, chain_offset)
        st = tr.sample(it)
        ctx.switch_partition()
        heads[it] = ctx.get_head(n_head)
        if trace_dir is not None:
            from .traces import TraceWriter
            vals = {k: np.asarray(v) for k, v in trace_func(heads[it], ctx.hamiltonian()[:, 0]).items()}
            if writer is None:
                writer = TraceWriter(trace_dir, B, n_iter, {k: v.shape[1:] for k, v in vals.items()})
            writer.write(it, vals)
        acc = _mean_over_all_chains(st["accept_stat"].sum(), B)
        acc_hist[it], eps_hist[it] = acc, float(np.mean(tr.step_size))
        nstep_hist[it], err_hist[it] = st["n_step"].mean(), st["integrator_error"].mean()
        if it >= n_adapt:
            outcome[:, 0] += st["moved"]
            outcome[:, 1] += ~st["moved"]
            outcome[:, 2] += st["integrator_error"]
            outcome[:, 3] += st["diverged"]
            outcome[:, 4] += st["n_step"] == 0
            outcome[:, 5] += ~st["h0_finite"]
            per_chain["n_step"] += st["n_step"]
            per_chain["accept"] += st["accept_stat"]
            per_chain["err"] += st["integrator_error"] | st["diverged"]
        if adapter is not None and it < n_adapt:
            if per_chain_step_size:
                tr.step_size = adapter.update(st["accept_stat"])  # [B]: every chain its own step size during warm-up
                if it == n_adapt - 1:
                    adapted = np.asarray(adapter.final(), dtype=np.float64).copy()  # per chain, before averaging
                    tr.step_size = _mean_over_all_chains(adapted.sum(), B)
            else:
                tr.step_size = adapter.update(acc)
                if it == n_adapt - 1:
                    tr.step_size = adapter.final()
        if callback is not None:
            callback(it, heads[it], acc, float(np.mean(tr.step_size)), st)
    n_main = max(n_iter - n_adapt, 1)
    out = dict(heads=heads, accept_stat=acc_hist, step_size=eps_hist, n_step=nstep_hist, integrator_error=err_hist,
               final_step_size=float(np.mean(tr.step_size)), adapted_step_sizes=adapted, chain_outcomes_dynamic=outcome,
               per_chain={k: v / n_main for k, v in per_chain.items()})
    if writer is not None:
        from .traces import save_summary
        writer.flush()
        c1 = ctx.counters()
        main = {k: np.asarray(v)[:, n_adapt:] for k, v in writer.arrays().items()}
        out["summary"] = save_summary(trace_dir, main, None, time.perf_counter() - t0, float(np.mean(tr.step_size)),
                                      {k: c1[k] - c0[k] for k in c1 if k != "_"})
        import os
        out["trace_files"] = {k: os.path.join(trace_dir, f"trace_{k}.npy") for k in writer.arrays()}
    return out
