"""ORACLE (test infrastructure only -- never imported by the product path).

CPU fp64 restatement of the reference's constrained system for conditioned diffusions,
operator for operator, with `torch.func` (jacrev / grad) standing in for JAX autodiff
and Python loops standing in for `lax.scan` / `lax.while_loop`.  Small sizes only.

PARITY UNPINNED: the reference has no tests or golden vectors for this path and its
third-party stack (jax 0.2.21 / mici 0.1.10 / symnum 0.1.2) is not installed here, so
this restatement is checked by mathematical invariants and independent derivations
(tests/test_oracle_py.py), not against outputs of the reference itself.

Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
import math
import numpy as onp
import torch
from torch.func import jacrev, grad

DT = torch.float64


def T(a):
    return torch.as_tensor(onp.asarray(a, dtype=onp.float64), dtype=DT)


def split(v, lengths):
    """sde/mici_extensions.py:31-40"""
    i = 0
    parts = []
    for j in lengths:
        parts.append(v[i:i + j])
        i += j
    if i < len(v):
        parts.append(v[i:])
    return parts


def split_and_reshape(array, shapes):
    """sde/mici_extensions.py:43-53"""
    i = 0
    parts = []
    for s in shapes:
        j = int(onp.prod(s))
        parts.append(array[i:i + j].reshape(tuple(s) + tuple(array.shape[1:])))
        i += j
    if i < array.shape[0]:
        parts.append(array[i:])
    return parts


def subseq_shapes(num_obs, num_steps_per_obs, num_obs_per_subseq):
    """Block shape tables, sde/mici_extensions.py:321-351."""
    if num_obs_per_subseq is None or num_obs_per_subseq == num_obs:
        return [((num_obs,),)], [((num_obs * num_steps_per_obs,),)], [(False,)]
    y_shapes, v_shapes, batched = [], [], []
    for init in [num_obs_per_subseq, num_obs_per_subseq // 2]:
        num_full, num_remaining = divmod(num_obs - init, num_obs_per_subseq)
        num_middle = num_full - 1 if num_remaining == 0 else num_full
        final = num_obs_per_subseq if num_remaining == 0 else num_remaining
        y_shapes.append(((init,),) + (((num_middle, num_obs_per_subseq),) if num_middle > 0 else ())
                        + ((final,),))
        v_shapes.append(((init * num_steps_per_obs,),)
                        + (((num_middle, num_obs_per_subseq * num_steps_per_obs),) if num_middle > 0 else ())
                        + ((final * num_steps_per_obs,),))
        batched.append((False, True, False) if num_middle > 0 else (False, False))
    return y_shapes, v_shapes, batched


class IdentityMetric:
    """Stand-in for mici.matrices.IdentityMatrix (only the identity metric is restated)."""


class ConditionedDiffusionConstrainedSystem:
    """sde/mici_extensions.py:208-1259 (identity metric only; M_0 != I is out of scope)."""

    def __init__(self, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, dim_u, dim_x, dim_v,
                 forward_func, generate_x_0, generate_z, obs_func, generate_σ=None,
                 use_gaussian_splitting=False, metric=None, dim_v_0=None):
        if metric is not None and not isinstance(metric, IdentityMetric):
            raise NotImplementedError("oracle restates the identity-metric paths only")
        self.use_gaussian_splitting = use_gaussian_splitting
        self.metric = IdentityMetric()
        y_seq = onp.asarray(y_seq, dtype=onp.float64)
        num_obs, dim_y = y_seq.shape
        δ = float(obs_interval) / num_steps_per_obs
        num_step = num_obs * num_steps_per_obs
        obs_indices = slice(num_steps_per_obs - 1, None, num_steps_per_obs)
        y_subseq_shapes, v_subseq_shapes, subseqs_are_batched = subseq_shapes(
            num_obs, num_steps_per_obs, num_obs_per_subseq)
        y_seq_t = T(y_seq)
        y_subseqs = [split_and_reshape(y_seq_t, shapes) for shapes in y_subseq_shapes]
        noisy_observations = generate_σ is not None
        if generate_σ is not None and isinstance(generate_σ, (int, float)):
            σ_const = float(generate_σ)

            def generate_σ(u):  # :354-358
                return torch.tensor(σ_const, dtype=DT)

        dim_v_0 = dim_x if dim_v_0 is None else dim_v_0
        self.y_subseqs = y_subseqs
        self.num_partition = len(y_subseqs)
        self.noisy_observations = noisy_observations
        self.dims = dict(dim_u=dim_u, dim_x=dim_x, dim_v=dim_v, dim_v_0=dim_v_0, dim_y=dim_y,
                         num_obs=num_obs, num_steps_per_obs=num_steps_per_obs)
        self.model_dict = {
            "dim_u": dim_u, "dim_v": dim_v, "dim_v_0": dim_v_0, "dim_y": dim_y, "num_obs": num_obs,
            "num_steps_per_obs": num_steps_per_obs, "δ": δ, "generate_z": generate_z,
            "generate_x_0": generate_x_0, "generate_σ": generate_σ, "forward_func": forward_func,
            "obs_func": obs_func, "y_seq": y_seq,
        }
        self.y_subseq_shapes = y_subseq_shapes
        self.v_subseq_shapes = v_subseq_shapes
        self.subseqs_are_batched = subseqs_are_batched

        def scan(z, x_0, v_seq):
            """lax.scan(lambda x, v: step_func(z, x, v), x_0, v_seq)  (:379-382, :396, :402)"""
            xs = []
            x = x_0
            for s in range(v_seq.shape[0]):
                x = forward_func(z, x, v_seq[s], δ)
                xs.append(x)
            return torch.stack(xs)

        def generate_x_obs_seq(q):  # :384-397
            if noisy_observations:
                u, v_0, v_seq_flat, _ = split(q, (dim_u, dim_v_0, num_obs * num_steps_per_obs * dim_v))
            else:
                u, v_0, v_seq_flat = split(q, (dim_u, dim_v_0))
            z = generate_z(u)
            x_0 = generate_x_0(z, v_0)
            v_seq = v_seq_flat.reshape((-1, dim_v))
            x_seq = scan(z, x_0, v_seq)
            return x_seq[obs_indices]

        def generate_y_bar(z, w_0, v_seq, σ_n_seq, initial_subseq, final_subseq):  # :399-411
            x_0 = generate_x_0(z, w_0) if initial_subseq else w_0
            x_seq = scan(z, x_0, v_seq)
            y_seq_ = obs_func(x_seq[obs_indices])
            if noisy_observations:
                y_seq_ = y_seq_ + σ_n_seq
            if final_subseq:
                return y_seq_.flatten()
            elif noisy_observations:
                return torch.cat((y_seq_.flatten(), x_seq[-1]))
            else:
                return torch.cat((y_seq_[:-1].flatten(), x_seq[-1]))

        def partition_into_subseqs(v_seq, v_0, n_seq, x_obs_seq, partition=0):  # :413-471
            end_y = None if noisy_observations else -1
            partition_size = len(y_subseq_shapes[partition])
            v_subseqs = split_and_reshape(v_seq, v_subseq_shapes[partition])
            if noisy_observations:
                n_subseqs = split_and_reshape(n_seq, y_subseq_shapes[partition])
            else:
                n_subseqs = (None,) * partition_size
            x_obs_subseqs = split_and_reshape(x_obs_seq, y_subseq_shapes[partition])
            w_inits = [v_0]
            prev_batched = False
            for b in range(1, partition_size):
                if subseqs_are_batched[partition][b]:
                    prev_last = x_obs_subseqs[b - 1][-1, -1] if prev_batched else x_obs_subseqs[b - 1][-1]
                    w_inits.append(torch.vstack([prev_last, x_obs_subseqs[b][:-1, -1]]))
                    prev_batched = True
                else:
                    # x_obs_subseqs[b-1][(-1,-1)] (batched predecessor) or [(-1,)] (:443-445)
                    w_inits.append(x_obs_subseqs[b - 1][-1, -1] if prev_batched else x_obs_subseqs[b - 1][-1])
                    prev_batched = False
            y_bars = []
            for b in range(0, partition_size - 1):
                ys = y_subseqs[partition][b]
                if subseqs_are_batched[partition][b]:
                    y_bars.append(torch.cat((ys[:, :end_y].reshape((ys.shape[0], -1)), x_obs_subseqs[b][:, -1]), -1))
                else:
                    y_bars.append(torch.cat((ys[:end_y].flatten(), x_obs_subseqs[b][-1])))
            y_bars.append(y_subseqs[partition][-1].flatten())
            return v_subseqs, n_subseqs, w_inits, y_bars

        def unpack(q):
            if noisy_observations:
                u, v_0, v_seq_flat, n_flat = split(q, (dim_u, dim_v_0, num_step * dim_v, num_obs * dim_y))
                n_seq = n_flat.reshape((-1, dim_y))
            else:
                u, v_0, v_seq_flat = split(q, (dim_u, dim_v_0))
                n_seq = None
            return u, v_0, v_seq_flat.reshape((-1, dim_v)), n_seq

        def constr(q, x_obs_seq, partition=0):  # :473-519
            u, v_0, v_seq, n_seq = unpack(q)
            z = generate_z(u)
            v_subseqs, n_subseqs, w_inits, y_bars = partition_into_subseqs(v_seq, v_0, n_seq, x_obs_seq, partition)
            partition_size = len(v_subseqs)
            if noisy_observations:
                σ = generate_σ(u)
                σ_n_subseqs = [σ * n_subseq for n_subseq in n_subseqs]
            else:
                σ_n_subseqs = (None,) * partition_size
            out = []
            for b in range(partition_size):
                first, last = b == 0, b == partition_size - 1
                if subseqs_are_batched[partition][b]:  # jax.vmap(generate_y_bar, ...) (:490-498)
                    vals = torch.stack([
                        generate_y_bar(z, w_inits[b][m], v_subseqs[b][m],
                                       σ_n_subseqs[b][m] if noisy_observations else None, first, last)
                        for m in range(v_subseqs[b].shape[0])])
                else:
                    vals = generate_y_bar(z, w_inits[b], v_subseqs[b], σ_n_subseqs[b], first, last)
                out.append((vals - y_bars[b]).flatten())
            return torch.cat(out)

        def jacob_constr_blocks(q, x_obs_seq, partition=0):  # :521-624
            def g_y_bar(u, v, n, w_0, initial_subseq, final_subseq):  # :559-569
                z = generate_z(u)
                if noisy_observations:
                    σ_n = generate_σ(u) * n
                else:
                    σ_n = None
                if initial_subseq:
                    w_0, v = split(v, (dim_v_0,))
                v_seq = v.reshape((-1, dim_v))
                return generate_y_bar(z, w_0, v_seq, σ_n, initial_subseq, final_subseq)

            u, v_0, v_seq, n_seq = unpack(q)
            v_subseqs, n_subseqs, w_inits, _ = partition_into_subseqs(v_seq, v_0, n_seq, x_obs_seq, partition)
            partition_size = len(v_subseqs)
            v_bars = [torch.cat([v_0, v_subseqs[0].flatten()])]
            for b in range(1, partition_size):
                v_bars.append(v_subseqs[b].reshape((v_subseqs[b].shape[0], -1))
                              if subseqs_are_batched[partition][b] else v_subseqs[b].flatten())
            dc_du_blocks, dc_dv_blocks = [], []
            for b in range(partition_size):
                first, last = b == 0, b == partition_size - 1

                def jac(v_bar, n, w_0):
                    f = lambda uu, vv: g_y_bar(uu, vv, n, w_0, first, last)
                    return jacrev(f, (0, 1))(u, v_bar)  # jax.jacrev(g_y_bar, (0, 1)) (:591)

                if subseqs_are_batched[partition][b]:
                    res = [jac(v_bars[b][m], n_subseqs[b][m] if noisy_observations else None, w_inits[b][m])
                           for m in range(v_bars[b].shape[0])]
                    dc_du_blocks.append(torch.stack([r[0] for r in res]))
                    dc_dv_blocks.append(torch.stack([r[1] for r in res]))
                else:
                    du, dv = jac(v_bars[b], n_subseqs[b], w_inits[b])
                    dc_du_blocks.append(du)
                    dc_dv_blocks.append(dv)
            if noisy_observations:  # :601-608
                σ = generate_σ(u)
                dc_dn_blocks = tuple(
                    (σ * torch.ones_like(n_subseqs[b])).reshape((n_subseqs[b].shape[0], -1) if is_batched else (-1,))
                    for b, is_batched in enumerate(subseqs_are_batched[partition]))
            else:
                dc_dn_blocks = (None,) * partition_size
            return tuple(dc_du_blocks), tuple(dc_dv_blocks), dc_dn_blocks

        def get_M_0_matrix():  # :794-798
            return torch.eye(dim_u, dtype=DT)

        def compute_D_blocks(dc_dv_l_blocks, dc_dn_l_blocks, dc_dv_r_blocks, dc_dn_r_blocks):  # :765-792
            D_blocks = [torch.einsum("...ij,...kj->...ik", l, r) for l, r in zip(dc_dv_l_blocks, dc_dv_r_blocks)]
            if noisy_observations:
                for b in range(len(D_blocks) - 1):
                    dn = dc_dn_l_blocks[b] * dc_dn_r_blocks[b]
                    add = torch.cat([dn, torch.zeros(D_blocks[b].shape[:-2] + (dim_x,), dtype=DT)], dim=-1)
                    D_blocks[b] = D_blocks[b] + torch.diag_embed(add)
                D_blocks[-1] = D_blocks[-1] + torch.diag_embed(dc_dn_l_blocks[-1] * dc_dn_r_blocks[-1])
            return D_blocks

        def chol_gram_blocks(dc_du_blocks, dc_dv_blocks, dc_dn_blocks):  # :626-687
            M_0 = get_M_0_matrix()
            D_blocks = compute_D_blocks(dc_dv_blocks, dc_dn_blocks, dc_dv_blocks, dc_dn_blocks)
            chol_D_blocks = tuple(torch.linalg.cholesky(D) for D in D_blocks)
            D_inv_dc_du_blocks = tuple(torch.cholesky_solve(du, ch) for ch, du in zip(chol_D_blocks, dc_du_blocks))
            acc = M_0
            for du, Dinv_du in zip(dc_du_blocks, D_inv_dc_du_blocks):
                acc = acc + (du.T @ Dinv_du if du.ndim == 2 else torch.einsum("ijk,ijl->kl", du, Dinv_du))
            chol_C = torch.linalg.cholesky(acc)
            return chol_C, chol_D_blocks

        def lu_jacob_product_blocks(dc_du_l, dc_dv_l, dc_dn_l, dc_du_r, dc_dv_r, dc_dn_r):  # :689-763
            M_0 = get_M_0_matrix()
            D_blocks = compute_D_blocks(dc_dv_l, dc_dn_l, dc_dv_r, dc_dn_r)
            lu_D = tuple(torch.linalg.lu_factor(D) for D in D_blocks)
            D_inv_du_l = tuple(torch.linalg.lu_solve(lu[0], lu[1], du) for lu, du in zip(lu_D, dc_du_l))
            acc = M_0
            for du_r, Dinv in zip(dc_du_r, D_inv_du_l):
                acc = acc + (du_r.T @ Dinv if du_r.ndim == 2 else torch.einsum("ijk,ijl->kl", du_r, Dinv))
            lu_C = torch.linalg.lu_factor(acc)
            return lu_C, lu_D

        def log_det_sqrt_gram_from_chol(chol_C, chol_D_blocks):  # :800-810
            return (sum(torch.log(torch.abs(torch.diagonal(ch, 0, -2, -1))).sum() for ch in chol_D_blocks)
                    + torch.log(torch.abs(torch.diagonal(chol_C))).sum() - 0.0)

        def log_det_sqrt_gram(q, x_obs_seq, partition=0):  # :812-820
            jac_blocks = jacob_constr_blocks(q, x_obs_seq, partition)
            chol_blocks = chol_gram_blocks(*jac_blocks)
            return log_det_sqrt_gram_from_chol(*chol_blocks), (jac_blocks, chol_blocks)

        def lmult_by_jacob_constr(dc_du_blocks, dc_dv_blocks, dc_dn_blocks, vct):  # :822-877
            if noisy_observations:
                vct_u, vct_v, vct_n = split(vct, (dim_u, dim_v_0 + num_obs * num_steps_per_obs * dim_v))
            else:
                vct_u, vct_v = split(vct, (dim_u,))
            vct_v_parts = split_and_reshape(
                vct_v, [(dv.shape[0], dv.shape[2]) if dv.ndim == 3 else (dv.shape[1],) for dv in dc_dv_blocks])
            dc_du = torch.vstack([du.reshape((-1, dim_u)) if du.ndim == 3 else du for du in dc_du_blocks])
            jacob_vct = dc_du @ vct_u + torch.cat([
                torch.einsum("ijk,ik->ij", dv, part).flatten() if dv.ndim == 3 else dv @ part
                for dv, part in zip(dc_dv_blocks, vct_v_parts)])
            if noisy_observations:
                vct_n_parts = split_and_reshape(vct_n, [tuple(dn.shape) for dn in dc_dn_blocks])
                pieces = []
                for dn, part in zip(dc_dn_blocks[:-1], vct_n_parts[:-1]):
                    if dn.ndim == 2:
                        pieces.append(torch.cat([dn * part, torch.zeros((dn.shape[0], dim_x), dtype=DT)], dim=1).flatten())
                    else:
                        pieces.append(torch.cat([dn * part, torch.zeros(dim_x, dtype=DT)]))
                pieces.append(dc_dn_blocks[-1] * vct_n_parts[-1])
                jacob_vct = jacob_vct + torch.cat(pieces)
            return jacob_vct

        def rmult_by_jacob_constr(dc_du_blocks, dc_dv_blocks, dc_dn_blocks, vct):  # :879-913
            vct_parts = split_and_reshape(vct, [tuple(du.shape[:-1]) for du in dc_du_blocks])
            out = [sum(torch.einsum("ij,ijk->k", p, du) if p.ndim == 2 else p @ du
                       for p, du in zip(vct_parts, dc_du_blocks))]
            out += [torch.einsum("ij,ijk->ik", p, dv).flatten() if p.ndim == 2 else p @ dv
                    for p, dv in zip(vct_parts, dc_dv_blocks)]
            if noisy_observations:
                out += [(p[:, :-dim_x] * dn).flatten() if p.ndim == 2 else p[:-dim_x] * dn
                        for p, dn in zip(vct_parts[:-1], dc_dn_blocks[:-1])]
                out += [vct_parts[-1] * dc_dn_blocks[-1]]
            return torch.cat(out)

        def cho_solve(ch, b):
            if b.ndim == ch.ndim - 1:
                return torch.cholesky_solve(b.unsqueeze(-1), ch).squeeze(-1)
            return torch.cholesky_solve(b, ch)

        def lu_solve(lu, b):
            if b.ndim == lu[0].ndim - 1:
                return torch.linalg.lu_solve(lu[0], lu[1], b.unsqueeze(-1)).squeeze(-1)
            return torch.linalg.lu_solve(lu[0], lu[1], b)

        def lmult_by_inv_gram(dc_du_blocks, dc_dv_blocks, dc_dn_blocks, chol_C, chol_D_blocks, vct):  # :915-942
            vct_parts = split_and_reshape(vct, [tuple(du.shape[:-1]) for du in dc_du_blocks])
            D_inv_vct = [cho_solve(ch, p) for ch, p in zip(chol_D_blocks, vct_parts)]
            s = sum(torch.einsum("...jk,...j->k", du, dv) for du, dv in zip(dc_du_blocks, D_inv_vct))
            y = cho_solve(chol_C, s)
            return torch.cat([cho_solve(ch, p - du @ y).flatten()
                              for ch, p, du in zip(chol_D_blocks, vct_parts, dc_du_blocks)])

        def lmult_by_inv_jacob_product(dc_du_l, dc_dv_l, dc_dn_l, dc_du_r, dc_dv_r, dc_dn_r, lu_C, lu_D, vct):  # :944-981
            vct_parts = split_and_reshape(vct, [tuple(du.shape[:-1]) for du in dc_du_l])
            D_inv_vct = [lu_solve(lu, p) for lu, p in zip(lu_D, vct_parts)]
            s = sum(torch.einsum("...jk,...j->k", du, dv) for du, dv in zip(dc_du_r, D_inv_vct))
            y = lu_solve(lu_C, s)
            return torch.cat([lu_solve(lu, p - du @ y).flatten() for lu, p, du in zip(lu_D, vct_parts, dc_du_l)])

        def normal_space_component(vct, jac_blocks, chol_blocks):  # :983-993
            return rmult_by_jacob_constr(
                *jac_blocks, lmult_by_inv_gram(*jac_blocks, *chol_blocks, lmult_by_jacob_constr(*jac_blocks, vct)))

        def norm(x):  # :995-997 (jnp.max propagates NaN)
            a = torch.abs(x)
            return torch.tensor(float("nan"), dtype=DT) if torch.isnan(a).any() else a.max()

        def _loop(q, body, constraint_tol, position_tol, divergence_tol, max_iters):
            """lax.while_loop(cond_func, body_func, (q, 0, 0, inf, -1.0)) (:1047-1059, :1119-1131)"""
            mu = torch.zeros_like(q)
            i, norm_delta_q, error = 0, float("inf"), -1.0
            while True:
                diverged = error > divergence_tol or math.isnan(error)
                converged = error < constraint_tol and norm_delta_q < position_tol
                if i >= max_iters or diverged or converged:
                    break
                q, mu, norm_delta_q, error = body(q, mu)
                i += 1
            return q, mu, i, norm_delta_q, error

        def quasi_newton_projection(q, x_obs_seq, partition, jac_prev, chol_prev, dt, constraint_tol,
                                    position_tol, divergence_tol, max_iters):  # :999-1063
            def body(q, mu):
                c = constr(q, x_obs_seq, partition)
                error = float(norm(c))
                delta_mu = rmult_by_jacob_constr(*jac_prev, lmult_by_inv_gram(*jac_prev, *chol_prev, c))
                delta_q = delta_mu
                return q - delta_q, mu + delta_mu, float(norm(delta_q)), error

            q, mu, i, ndq, err = _loop(q, body, constraint_tol, position_tol, divergence_tol, max_iters)
            return q, (mu / math.sin(dt) if use_gaussian_splitting else mu / dt), i, ndq, err

        def newton_projection(q, x_obs_seq, partition, jac_prev, dt, constraint_tol, position_tol,
                              divergence_tol, max_iters):  # :1065-1135
            def body(q, mu):
                c = constr(q, x_obs_seq, partition)
                jac_curr = jacob_constr_blocks(q, x_obs_seq, partition)
                lus = lu_jacob_product_blocks(*jac_curr, *jac_prev)
                error = float(norm(c))
                delta_mu = rmult_by_jacob_constr(
                    *jac_prev, lmult_by_inv_jacob_product(*jac_curr, *jac_prev, *lus, c))
                delta_q = delta_mu
                return q - delta_q, mu + delta_mu, float(norm(delta_q)), error

            q, mu, i, ndq, err = _loop(q, body, constraint_tol, position_tol, divergence_tol, max_iters)
            return q, (mu / math.sin(dt) if use_gaussian_splitting else mu / dt), i, ndq, err

        self._generate_x_obs_seq = generate_x_obs_seq
        self._constr = constr
        self._jacob_constr_blocks = jacob_constr_blocks
        self._chol_gram_blocks = chol_gram_blocks
        self._lu_jacob_product_blocks = lu_jacob_product_blocks
        self._compute_D_blocks = compute_D_blocks
        self._log_det_sqrt_gram_from_chol = log_det_sqrt_gram_from_chol
        self._log_det_sqrt_gram = log_det_sqrt_gram

        def value_and_grad_log_det(q, x_obs_seq, partition):  # jax.value_and_grad(..., has_aux=True) (:1143-1146)
            def f(qq):
                val, ((du, dv, dn), chol) = log_det_sqrt_gram(qq, x_obs_seq, partition)
                if not noisy_observations:  # torch.func aux outputs must be tensors
                    dn = ()
                return val, (val, ((du, dv, dn), chol))

            g, (val, ((du, dv, dn), chol)) = grad(f, has_aux=True)(q)
            if not noisy_observations:
                dn = (None,) * len(du)
            return (val, ((du, dv, dn), chol)), g

        self._grad_log_det_sqrt_gram = value_and_grad_log_det
        self._lmult_by_jacob_constr = lmult_by_jacob_constr
        self._rmult_by_jacob_constr = rmult_by_jacob_constr
        self._lmult_by_inv_gram = lmult_by_inv_gram
        self._lmult_by_inv_jacob_product = lmult_by_inv_jacob_product
        self._normal_space_component = normal_space_component
        self._quasi_newton_projection = quasi_newton_projection
        self._newton_projection = newton_projection

    # ---- Mici System surface (:1151-1259); `state` is a ConditionedDiffusionHamiltonianState
    def _cached(self, state, key, fn):
        ck = (key, state._version)
        if state._cache.get("key_" + key) != ck:
            state._cache[key] = fn()
            state._cache["key_" + key] = ck
            state._call_counts[key] = state._call_counts.get(key, 0) + 1
        return state._cache[key]

    def constr(self, state):  # :1151-1155
        return self._cached(state, "constr", lambda: self._constr(T(state.pos), T(state.x_obs_seq), state.partition).numpy())

    def jacob_constr_blocks(self, state):  # :1157-1161
        return self._cached(state, "jacob_constr_blocks",
                            lambda: self._jacob_constr_blocks(T(state.pos), T(state.x_obs_seq), state.partition))

    def chol_gram_blocks(self, state):  # :1163-1167
        return self._cached(state, "chol_gram_blocks", lambda: self._chol_gram_blocks(*self.jacob_constr_blocks(state)))

    def log_det_sqrt_gram(self, state):  # :1169-1171
        return self._cached(state, "log_det_sqrt_gram",
                            lambda: float(self._log_det_sqrt_gram_from_chol(*self.chol_gram_blocks(state))))

    def grad_log_det_sqrt_gram(self, state):  # :1173-1184 (fills the three aux caches as well)
        def run():
            (val, (jac, chol)), g = self._grad_log_det_sqrt_gram(T(state.pos), T(state.x_obs_seq), state.partition)
            for k, v in (("log_det_sqrt_gram", float(val)), ("jacob_constr_blocks", _detach(jac)),
                         ("chol_gram_blocks", _detach(chol))):
                state._cache[k] = v
                state._cache["key_" + k] = (k, state._version)
            return g.detach().numpy()

        return self._cached(state, "grad_log_det_sqrt_gram", run)

    def neg_log_dens(self, state):  # :56-58
        return 0.5 * float(onp.sum(state.pos ** 2))

    def grad_neg_log_dens(self, state):  # :61-63
        return state.pos

    def h1(self, state):  # :1186-1190
        if self.use_gaussian_splitting:
            return self.log_det_sqrt_gram(state)
        return self.neg_log_dens(state) + self.log_det_sqrt_gram(state)

    def dh1_dpos(self, state):  # :1192-1196
        if self.use_gaussian_splitting:
            return self.grad_log_det_sqrt_gram(state)
        return self.grad_neg_log_dens(state) + self.grad_log_det_sqrt_gram(state)

    def h2(self, state):  # :1198-1202
        if self.use_gaussian_splitting:
            return 0.5 * state.pos @ state.pos + 0.5 * state.mom @ state.mom
        return 0.5 * state.mom @ state.mom

    def h(self, state):
        return self.h1(state) + self.h2(state)

    def dh2_dmom(self, state):  # :1204-1208
        return state.mom

    def h1_flow(self, state, dt):  # mici System.h1_flow: mom -= dt * dh1_dpos
        state.mom = state.mom - dt * self.dh1_dpos(state)

    def h2_flow(self, state, dt):  # :1222-1231
        if self.use_gaussian_splitting:
            sin_dt, cos_dt = onp.sin(dt), onp.cos(dt)
            pos = state.pos.copy()
            state.pos = state.pos * cos_dt + sin_dt * state.mom
            state.mom = state.mom * cos_dt - sin_dt * pos
        else:
            state.pos = state.pos + dt * self.dh2_dmom(state)

    def dh2_flow_dmom(self, dt):  # :1233-1238 (scalars standing for scalar * IdentityMatrix)
        if self.use_gaussian_splitting:
            return onp.sin(dt), onp.cos(dt)
        return dt, 1.0

    def update_x_obs_seq(self, state):  # :1240-1241
        state.x_obs_seq = self._generate_x_obs_seq(T(state.pos)).numpy()

    def normal_space_component(self, state, vct):  # :1243-1250
        return self._normal_space_component(T(vct), self.jacob_constr_blocks(state),
                                            self.chol_gram_blocks(state)).numpy()

    def project_onto_cotangent_space(self, mom, state):  # :1252-1254
        return mom - self.normal_space_component(state, mom)

    def sample_momentum(self, state, rng):  # :1256-1259
        mom = rng.standard_normal(state.pos.shape)
        return self.project_onto_cotangent_space(mom, state)


def _detach(tree):
    if isinstance(tree, torch.Tensor):
        return tree.detach()
    if isinstance(tree, (tuple, list)):
        return tuple(_detach(t) for t in tree)
    return tree


class ConvergenceError(RuntimeError):
    """mici.errors.ConvergenceError"""


class NonReversibleStepError(RuntimeError):
    """mici.errors.NonReversibleStepError"""


class ConditionedDiffusionHamiltonianState:
    """sde/mici_extensions.py:1285-1320 (+ the parts of mici.states.ChainState it relies on)."""

    def __init__(self, pos, x_obs_seq, partition=0, mom=None, dir=1, _call_counts=None):
        self.__dict__["_version"] = 0
        self.__dict__["_cache"] = {}
        self.__dict__["_call_counts"] = {} if _call_counts is None else _call_counts
        self.pos = onp.array(pos, dtype=onp.float64)
        self.x_obs_seq = onp.array(x_obs_seq, dtype=onp.float64)
        self.partition = int(partition)
        self.mom = None if mom is None else onp.array(mom, dtype=onp.float64)
        self.dir = dir

    def __setattr__(self, k, v):
        if k in ("pos", "x_obs_seq", "partition"):  # cache dependencies (:1151-1176)
            self.__dict__["_version"] = self.__dict__["_version"] + 1
            self.__dict__["_cache"] = {}
        self.__dict__[k] = v

    def copy(self):
        new = ConditionedDiffusionHamiltonianState.__new__(ConditionedDiffusionHamiltonianState)
        new.__dict__.update(
            _version=self._version, _cache=dict(self._cache), _call_counts=self._call_counts,
            pos=self.pos.copy(), x_obs_seq=self.x_obs_seq.copy(), partition=self.partition,
            mom=None if self.mom is None else self.mom.copy(), dir=self.dir)
        return new


def _finish_projection(name, state, q_, mu, i, norm_delta_q, error, dh2_flow_mom_dmom, constraint_tol,
                       position_tol, divergence_tol):
    """Acceptance test / error mapping shared by both wrappers (:1388-1402, :1462-1476)."""
    if error < constraint_tol and norm_delta_q < position_tol:
        state.pos = q_.numpy().copy()
        if state.mom is not None:
            state.mom = state.mom - dh2_flow_mom_dmom * mu.numpy()
        return state
    elif error > divergence_tol or onp.isnan(error):
        raise ConvergenceError(f"{name} iteration diverged on iteration {i}. Last |c|={error:.1e}, |δq|={norm_delta_q}.")
    else:
        raise ConvergenceError(f"{name} iteration did not converge. Last |c|={error:.1e}, |δq|={norm_delta_q}.")


def jitted_solve_projection_onto_manifold_quasi_newton(state, state_prev, dt, system, constraint_tol=1e-8,
                                                       position_tol=1e-8, divergence_tol=1e10, max_iters=50):
    """sde/mici_extensions.py:1323-1402"""
    jac_prev = system.jacob_constr_blocks(state_prev)
    chol_prev = system.chol_gram_blocks(state_prev)
    _, dh2_flow_mom_dmom = system.dh2_flow_dmom(dt)
    q_, mu, i, ndq, err = system._quasi_newton_projection(
        T(state.pos), T(state.x_obs_seq), state.partition, jac_prev, chol_prev, dt, constraint_tol, position_tol,
        divergence_tol, max_iters)
    state._call_counts["constr"] = state._call_counts.get("constr", 0) + i
    state.last_iters = i
    return _finish_projection("Quasi-Newton", state, q_, mu, i, ndq, err, dh2_flow_mom_dmom, constraint_tol,
                              position_tol, divergence_tol)


def jitted_solve_projection_onto_manifold_newton(state, state_prev, dt, system, constraint_tol=1e-8,
                                                 position_tol=1e-8, divergence_tol=1e10, max_iters=50):
    """sde/mici_extensions.py:1405-1476"""
    jac_prev = system.jacob_constr_blocks(state_prev)
    _, dh2_flow_mom_dmom = system.dh2_flow_dmom(dt)
    q_, mu, i, ndq, err = system._newton_projection(
        T(state.pos), T(state.x_obs_seq), state.partition, jac_prev, dt, constraint_tol, position_tol,
        divergence_tol, max_iters)
    for k in ("constr", "jacob_constr_blocks", "lu_jacob_product_blocks"):
        state._call_counts[k] = state._call_counts.get(k, 0) + i
    state.last_iters = i
    return _finish_projection("Newton", state, q_, mu, i, ndq, err, dh2_flow_mom_dmom, constraint_tol,
                              position_tol, divergence_tol)


class ConstrainedLeapfrogIntegrator:
    """mici 0.1.10 `integrators.ConstrainedLeapfrogIntegrator` (NOT in /root/reference; restated from the
    published algorithm; call site scripts/utils.py:284-290)."""

    def __init__(self, system, step_size=None, n_inner_step=1, reverse_check_tol=2e-8,
                 projection_solver=jitted_solve_projection_onto_manifold_newton, projection_solver_kwargs=None):
        self.system = system
        self.step_size = step_size
        self.n_inner_step = n_inner_step
        self.reverse_check_tol = reverse_check_tol
        self.projection_solver = projection_solver
        self.projection_solver_kwargs = projection_solver_kwargs or {}

    def _h2_flow_retraction_onto_manifold(self, state, state_prev, dt):
        self.system.h2_flow(state, dt)
        self.projection_solver(state, state_prev, dt, self.system, **self.projection_solver_kwargs)

    def _project_onto_cotangent_space(self, state):
        state.mom = self.system.project_onto_cotangent_space(state.mom, state)

    def _step_a(self, state, dt):
        self.system.h1_flow(state, dt)
        self._project_onto_cotangent_space(state)

    def _step_b(self, state, dt):
        dt_i = dt / self.n_inner_step
        for i in range(self.n_inner_step):
            state_prev = state.copy()
            self._h2_flow_retraction_onto_manifold(state, state_prev, dt_i)
            if i == self.n_inner_step - 1:
                self.system.dh1_dpos(state)  # pre-evaluate; fills J / chol / log-det caches
            self._project_onto_cotangent_space(state)
            state_back = state.copy()
            self._h2_flow_retraction_onto_manifold(state_back, state, -dt_i)
            rev_diff = onp.max(onp.abs(state_back.pos - state_prev.pos))  # mici.solvers.maximum_norm
            self.last_rev_diff = float(rev_diff)
            self.last_iters = (getattr(state, "last_iters", -1), getattr(state_back, "last_iters", -1))
            if rev_diff > self.reverse_check_tol:
                raise NonReversibleStepError(
                    f"Non-reversible step. Distance between initial and forward-backward integrated positions = {rev_diff:.1e}.")

    def step(self, state):
        state = state.copy()
        dt = state.dir * self.step_size
        self._step_a(state, 0.5 * dt)
        self._step_b(state, dt)
        self._step_a(state, 0.5 * dt)
        return state


class SwitchPartitionTransition:
    """sde/mici_extensions.py:1262-1282"""

    def __init__(self, system):
        self.system = system
        self.num_partition = system.num_partition

    def sample(self, state, rng=None):
        state.partition = (state.partition + 1) % self.num_partition
        self.system.update_x_obs_seq(state)
        return state, None


def find_initial_state_by_linear_interpolation(system, rng, generate_x_obs_seq_init, u=None, v_0=None):
    """sde/mici_extensions.py:1479-1547"""
    md = system.model_dict
    δ, S, dim_v = md["δ"], md["num_steps_per_obs"], md["dim_v"]

    def solve_inner(z, x, Δx):  # :1495-1511
        f0 = md["forward_func"](z, x, torch.zeros(dim_v, dtype=DT), δ) - x
        A = jacrev(lambda v: md["forward_func"](z, x, v, δ) - x)(torch.zeros(dim_v, dtype=DT))
        return torch.linalg.lstsq(A, (Δx - f0).unsqueeze(-1)).solution.squeeze(-1)

    u = rng.standard_normal(md["dim_u"]) if u is None else u
    z = md["generate_z"](T(u))
    v_0 = rng.standard_normal(md["dim_v_0"]) if v_0 is None else v_0
    x_0 = md["generate_x_0"](z, T(v_0))
    x_obs_seq = onp.asarray(generate_x_obs_seq_init(rng))
    x_obs_t = T(x_obs_seq)
    x_0_seq = torch.cat((x_0[None], x_obs_t[:-1]))  # :1521
    v_rows = []
    for x_a, x_b in zip(x_0_seq, x_obs_t):
        Δx = (x_b - x_a) / S
        for s in range(S):
            v_rows.append(solve_inner(z, x_a + s * Δx, Δx))
    v_seq = torch.stack(v_rows).numpy()
    if system.noisy_observations:
        q = onp.concatenate([u, v_0, v_seq.flatten(), onp.zeros(md["dim_y"] * md["num_obs"])])
    else:
        q = onp.concatenate([u, v_0, v_seq.flatten()])
    state = ConditionedDiffusionHamiltonianState(pos=q, x_obs_seq=x_obs_seq)
    state.mom = system.sample_momentum(state, rng)
    return state


def make_system(model, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, sigma=None,
                use_gaussian_splitting=False):
    """Wiring of scripts/utils.py:254-270 for a model object of oracle.py.models.  sigma: None, a number, or "variable"
    = the model's generate_σ_y with dim_u = dim_z + 1 (scripts/sir_model_chmc_experiment.py:44,58,77)."""
    variable = isinstance(sigma, str)
    return ConditionedDiffusionConstrainedSystem(
        obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, model.dim_z + int(variable), model.dim_x, model.dim_v,
        model.forward_func, model.generate_x_0, model.generate_z, model.obs_func,
        generate_σ=model.generate_sigma_y if variable else sigma,
        use_gaussian_splitting=use_gaussian_splitting, dim_v_0=model.dim_v_0)
