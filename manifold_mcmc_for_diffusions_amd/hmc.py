"""Batched unconstrained HMC on the reference's comparator target (scripts/*_hmc_experiment.py, scripts/utils.py:203-250:
`mici.systems.EuclideanMetricSystem(neg_log_dens, grad_neg_log_dens)` on `conditioned_diffusion_neg_log_dens_and_grad`,
sde/mici_extensions.py:82-205, with the noise of the observations marginalised instead of constrained).

All chains of a context advance together; positions, momenta and gradients are torch tensors in device memory and the
target is evaluated by the library on those buffers (`chmc_neg_log_dens_and_grad_device`).  Static (optionally
jittered) leapfrog trajectories with a Metropolis accept step, dual-averaging step-size adaptation on the mean accept
statistic of all chains, and the metric options of the reference's script: "identity", "diagonal" (online variances) and
"block" (`OnlineBlockDiagonalMetricAdapter(dim_u + dim_v_0)`: dense block for the global parameters and the initial
state, identity for the noise increments)."""
import numpy as np
from .sampling import DualAveragingStepSize, _mean_over_all_chains
from .adapters import OnlineBlockDiagonalMetricAdapter


class _Metric:
    """M = blockdiag(M_0, I) or diag(m): sqrt @ n, inv @ p on [B, Q] tensors."""

    def __init__(self, torch, Q, device):
        self.torch, self.Q, self.dev = torch, Q, device
        self.kind, self.L, self.W, self.d = "identity", None, None, None

    def set_block(self, M0):
        t = self.torch
        self.kind = "block"
        self.L = t.from_numpy(np.linalg.cholesky(M0)).to(self.dev)
        self.W = t.from_numpy(np.linalg.inv(M0)).to(self.dev)

    def set_diag(self, var):
        self.kind = "diagonal"
        self.d = self.torch.from_numpy(1.0 / np.asarray(var)).to(self.dev)  # metric = inverse of the variance estimates

    def sample(self, n):
        if self.kind == "block":
            k = self.L.shape[0]
            n[:, :k] = n[:, :k] @ self.L.T
        elif self.kind == "diagonal":
            n *= self.d.sqrt()
        return n

    def inv(self, p):
        if self.kind == "block":
            v = p.clone()
            k = self.W.shape[0]
            v[:, :k] = p[:, :k] @ self.W.T
            return v
        if self.kind == "diagonal":
            return p / self.d
        return p


def sample_hmc(ctx, q_init, n_iter, n_step, step_size, seed, n_adapt=0, metric_type="identity", jitter_length=False,
               use_gaussian_splitting=False, n_head=6, callback=None, metric_window=0.75, metric_skip=0.25):
    """q_init [B, U + V0 + T S V] (no observation-noise components).  Returns traces of the first n_head components,
    accept statistics, step sizes and the adapted metric."""
    import torch
    dev = torch.device("cuda", 0) if ctx.L.chmc_backend().startswith(b"hip") else torch.device("cpu")
    B, QH = ctx.B, ctx.U + ctx.NV
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    rng = np.random.default_rng(seed)
    q = torch.from_numpy(np.ascontiguousarray(q_init, dtype=np.float64)).to(dev)
    g, qn, gn = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    metric = _Metric(torch, QH, dev)

    def target(x, grad):
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return ctx.neg_log_dens_and_grad_device(x.data_ptr(), grad.data_ptr(), use_gaussian_splitting)

    u = target(q, g)
    adapter = DualAveragingStepSize(step_size) if n_adapt > 0 else None
    dim_param = ctx.U + ctx.V0
    blk = OnlineBlockDiagonalMetricAdapter(dim_param) if metric_type == "block" else None
    n_met = max(2, int(metric_window * n_adapt)) if metric_type != "identity" and n_adapt > 0 else 0
    n_skip = min(int(metric_skip * n_adapt), max(n_met - 2, 0))
    blk_state, w_n, w_mean, w_m2 = None, 0, None, None
    heads = np.empty((n_iter, B, n_head))
    acc_hist, eps_hist = np.empty(n_iter), np.empty(n_iter)
    for it in range(n_iter):
        p = metric.sample(torch.randn(B, QH, dtype=torch.float64, device=dev, generator=gen))
        h0 = u + 0.5 * (p * metric.inv(p)).sum(1).cpu().numpy()
        length = 1 + np.floor(rng.random(B) * n_step).astype(np.int64) if jitter_length else np.full(B, n_step)
        qn.copy_(q), gn.copy_(g)
        un = u.copy()
        for k in range(int(length.max())):
            live = torch.from_numpy(k < length).to(dev)[:, None]
            p = torch.where(live, p - 0.5 * step_size * gn, p)
            qn = torch.where(live, qn + step_size * metric.inv(p), qn)
            gl = torch.empty_like(gn)
            ul = target(qn, gl)
            lv = k < length
            un = np.where(lv, ul, un)
            gn = torch.where(live, gl, gn)
            p = torch.where(live, p - 0.5 * step_size * gn, p)
        h1 = un + 0.5 * (p * metric.inv(p)).sum(1).cpu().numpy()
        dh = h1 - h0
        prob = np.where(np.isfinite(dh), np.exp(np.minimum(0.0, -np.where(np.isfinite(dh), dh, np.inf))), 0.0)
        accept = rng.random(B) < prob
        am = torch.from_numpy(accept).to(dev)[:, None]
        q = torch.where(am, qn, q)
        g = torch.where(am, gn, g)
        u = np.where(accept, un, u)
        heads[it] = q[:, :n_head].cpu().numpy()
        if n_met and n_skip <= it < n_met:
            if blk is not None:
                pos = q[:, :dim_param].cpu().numpy()
                if blk_state is None:
                    blk_state = blk.initialize(np.zeros((B, QH)))
                blk.update(blk_state, pos)
            else:  # online variances pooled over chains (Welford on the per-draw chain means and squares)
                x = q
                w_n += 1
                if w_mean is None:
                    w_mean, w_m2 = torch.zeros_like(x), torch.zeros_like(x)
                d0 = x - w_mean
                w_mean += d0 / w_n
                w_m2 += d0 * (x - w_mean)
            if it == n_met - 1:
                if blk is not None:
                    metric.set_block(blk.finalize(blk_state).blocks[0].array)
                else:
                    n_tot = w_n * B
                    var = (w_m2.sum(0) + w_n * ((w_mean - w_mean.mean(0)) ** 2).sum(0)) / max(n_tot - 1, 1)
                    var = var * n_tot / (5 + n_tot) + 1e-3 * 5 / (5 + n_tot)  # regularisation as in Mici / Stan
                    metric.set_diag(var.cpu().numpy())
                adapter = DualAveragingStepSize(step_size)
        acc_all = _mean_over_all_chains(prob.sum(), B)
        acc_hist[it], eps_hist[it] = acc_all, step_size
        if adapter is not None and it < n_adapt:
            step_size = adapter.update(acc_all)
            if it == n_adapt - 1:
                step_size = adapter.final()
        if callback is not None:
            callback(it, heads[it], acc_all, step_size)
    return dict(heads=heads, accept_stat=acc_hist, step_size=eps_hist, final_step_size=step_size, metric=metric,
                final_q=q.cpu().numpy())
