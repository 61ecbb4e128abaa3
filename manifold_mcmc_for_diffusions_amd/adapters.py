"""Warm-up adapters of the reference that live next to the system class.

OnlineBlockDiagonalMetricAdapter restates sde/mici_extensions.py:1804-1931 for batched chains: every chain keeps its
own Welford statistics of the first `dim_param` position components, and `finalize` combines the per-chain statistics
(of all ranks, when `torch.distributed` is initialised) with the parallel covariance update of Schubert and Gertz, in
chain order, exactly as the reference combines the adapter states of its independent chains."""
import numpy as np
from .errors import AdaptationError
from .system import DensePositiveDefiniteMatrix, IdentityMatrix, PositiveDefiniteBlockDiagonalMatrix


class OnlineBlockDiagonalMetricAdapter:
    is_fast = False

    def __init__(self, dim_param, reg_iter_offset=5, reg_scale=1e-3):  # :1842-1855
        self.dim_param = dim_param
        self.reg_iter_offset = reg_iter_offset
        self.reg_scale = reg_scale

    def initialize(self, pos):  # :1857-1866, one state per chain: pos [B, Q] (or [Q])
        pos = np.atleast_2d(np.asarray(pos))
        B, d = pos.shape[0], self.dim_param
        return {"iter": np.zeros(B, dtype=np.int64), "mean": np.zeros((B, d)), "sum_diff_outer": np.zeros((B, d, d)),
                "dim_pos": pos.shape[1]}

    def update(self, adapt_state, pos, mask=None):  # :1868-1879 (Welford); mask [B]: chains that contribute this draw
        x = np.atleast_2d(np.asarray(pos))[:, :self.dim_param]
        m = np.ones(x.shape[0], dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        adapt_state["iter"][m] += 1
        d0 = x - adapt_state["mean"]
        new_mean = adapt_state["mean"] + d0 / np.maximum(adapt_state["iter"], 1)[:, None]
        adapt_state["mean"][m] = new_mean[m]
        d1 = x - adapt_state["mean"]
        # pos_minus_mean[None, :] * (pos - mean)[:, None]: entry [i, j] = d1_i d0_j
        adapt_state["sum_diff_outer"][m] += (d1[:, :, None] * d0[:, None, :])[m]

    def _regularize_covar_est(self, covar_est, n_iter):  # :1881-1890
        covar_est *= n_iter / (self.reg_iter_offset + n_iter)
        covar_est[np.diag_indices_from(covar_est)] += self.reg_scale * (self.reg_iter_offset / (self.reg_iter_offset + n_iter))

    @staticmethod
    def combine(iters, means, sums):
        """Schubert and Gertz (2018) parallel covariance combination of per-chain statistics, in order (:1899-1918)."""
        n_iter, mean_est, covar_est = 0, None, None
        for k in range(len(iters)):
            if mean_est is None:
                n_iter, mean_est, covar_est = int(iters[k]), means[k].copy(), sums[k].copy()
                continue
            n_prev = n_iter
            n_iter += int(iters[k])
            if n_iter == 0:
                continue
            mean_diff = mean_est - means[k]
            mean_est = (mean_est * n_prev + iters[k] * means[k]) / n_iter
            covar_est = covar_est + sums[k] + np.outer(mean_diff, mean_diff) * (iters[k] * n_prev) / n_iter
        return n_iter, mean_est, covar_est

    def finalize(self, adapt_state, system=None):  # :1892-1931
        """Returns the metric; assigns it to `system.metric` when a system is given."""
        iters, means, sums = adapt_state["iter"], adapt_state["mean"], adapt_state["sum_diff_outer"]
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                parts = [None] * dist.get_world_size()
                dist.all_gather_object(parts, (iters, means, sums))  # rank order = global chain order
                iters = np.concatenate([p[0] for p in parts])
                means = np.concatenate([p[1] for p in parts])
                sums = np.concatenate([p[2] for p in parts])
        except ImportError:
            pass
        n_iter, _, covar_est = self.combine(iters, means, sums)
        if n_iter < 2:
            raise AdaptationError("At least two chain samples required to compute a variance estimates.")
        covar_est = covar_est / (n_iter - 1)
        self._regularize_covar_est(covar_est, n_iter)
        metric = PositiveDefiniteBlockDiagonalMatrix(
            (DensePositiveDefiniteMatrix(covar_est).inv, IdentityMatrix(adapt_state["dim_pos"] - self.dim_param)))
        if system is not None:
            system.metric = metric
        return metric
