"""Initial chain states on the constraint manifold (host side, NumPy, vectorised over time steps).

`find_initial_state_by_linear_interpolation` restates sde/mici_extensions.py:1479-1547: draw parameters and the
initial state from their priors, pick a full-state sequence at the observation times consistent with the data
and solve, step by step, for the noise vectors that make the discretised path interpolate linearly between
those states (the one-step map is affine in `v` with a full-rank `d forward_func / d v`).
"""
import numpy as np


def solve_for_v_seq(model, z, x_0, x_obs_seq, num_steps_per_obs, δ):
    """solve_for_v_seq (:1503-1526): returns v_seq [T*S, V]."""
    S = num_steps_per_obs
    x_a = np.concatenate((x_0[None], x_obs_seq[:-1]))  # :1521
    Δ = (x_obs_seq - x_a) / S  # :1514
    x_s = x_a[:, None, :] + np.arange(S)[None, :, None] * Δ[:, None, :]  # :1515-1518  [T, S, X]
    zero_v = np.zeros(x_s.shape[:-1] + (model.dim_v,))
    mean_diff = model._forward(z, x_s, zero_v, δ) - x_s  # :1495-1501
    A = model.noise_matrix(z, x_s, δ)
    rhs = np.broadcast_to(Δ[:, None, :], x_s.shape) - mean_diff
    if model.dim_v == model.dim_x:
        v = np.linalg.solve(A, rhs[..., None])[..., 0]
    else:  # np.linalg.lstsq per step (:1511)
        v = np.stack([np.linalg.lstsq(a, r, rcond=None)[0] for a, r in
                      zip(A.reshape((-1,) + A.shape[-2:]), rhs.reshape((-1, rhs.shape[-1])))]).reshape(
                          x_s.shape[:-1] + (model.dim_v,))
    return v.reshape((-1, model.dim_v))


def find_initial_state_by_linear_interpolation(model, obs_interval, num_steps_per_obs, y_seq, rng,
                                               generate_x_obs_seq_init, noisy, u=None, v_0=None, dim_u=None):
    """Returns (q, x_obs_seq) for one chain; RNG draw order follows the reference (:1528-1532).
    dim_u: dim_z + 1 with variable observation noise (default dim_z)."""
    δ = obs_interval / num_steps_per_obs
    u = rng.standard_normal(model.dim_z if dim_u is None else dim_u) if u is None else np.asarray(u, dtype=np.float64)
    z = model.generate_z(u)
    v_0 = rng.standard_normal(model.dim_v_0) if v_0 is None else np.asarray(v_0, dtype=np.float64)
    x_0 = model.generate_x_0(z, v_0)
    x_obs_seq = np.asarray(generate_x_obs_seq_init(rng), dtype=np.float64)
    v_seq = solve_for_v_seq(model, z, x_0, x_obs_seq, num_steps_per_obs, δ)
    parts = [u, v_0, v_seq.flatten()]
    if noisy:
        parts.append(np.zeros(np.asarray(y_seq).size))  # :1538-1539
    return np.concatenate(parts), x_obs_seq


def fhn_initial_draws(model, y_seq, num_chains, seed=20200710, chain_offset=0, total_chains=None):
    """The random inputs of the FitzHugh-Nagumo initial states (scripts/fhn_model_noisy_obs_chmc_experiment.py:105-117),
    one independent generator per chain (SeedSequence(seed).spawn(total)[chain]) so that any sharding of the chains
    over ranks gives the same draws: u [B, Z], v_0 [B, V0], x_obs_seq_init [B, T, X] = [y, 0.5 N(0, 1)], generators."""
    y_seq = np.asarray(y_seq, dtype=np.float64).reshape((-1, 1))
    total = num_chains + chain_offset if total_chains is None else total_chains
    seqs = np.random.SeedSequence(seed).spawn(total)[chain_offset:chain_offset + num_chains]
    rngs = [np.random.default_rng(s) for s in seqs]
    us, v0s, xos = [], [], []
    for rng in rngs:
        us.append(rng.standard_normal(model.dim_z))
        v0s.append(rng.standard_normal(model.dim_v))  # sic: the script draws dim_v, equal to dim_v_0 for this model (:112)
        xos.append(np.concatenate((y_seq, rng.standard_normal(y_seq.shape) * 0.5), -1))  # :105-106
    return np.stack(us), np.stack(v0s), np.stack(xos), rngs


def fhn_initial_states(model, obs_interval, num_steps_per_obs, y_seq, num_chains, noisy, seed=20200710,
                       chain_offset=0, total_chains=None):
    """Initial states for `num_chains` FitzHugh-Nagumo chains on the host (NumPy).
    Returns q [B, Q], x_obs_seq [B, T, X], and the per-chain generators (next draw: the momentum)."""
    us, v0s, xos, rngs = fhn_initial_draws(model, y_seq, num_chains, seed, chain_offset, total_chains)
    qs = []
    for u, v_0, xo in zip(us, v0s, xos):
        q, _ = find_initial_state_by_linear_interpolation(model, obs_interval, num_steps_per_obs, y_seq, None,
                                                          lambda rng, xo=xo: xo, noisy, u=u, v_0=v_0)
        qs.append(q)
    return np.stack(qs), xos, rngs


def fhn_initial_states_device(ctx, model, y_seq, seed=20200710, chain_offset=0, total_chains=None, partition=0):
    """The same initial states solved on the device for all chains at once (`chmc_init_linear_interpolation`); the
    state stays resident, nothing but the O(B T) draws crosses PCIe.  Returns the per-chain generators."""
    us, v0s, xos, rngs = fhn_initial_draws(model, y_seq, ctx.B, seed, chain_offset, total_chains)
    ctx.init_by_linear_interpolation(us, v0s, xos, partition)
    return rngs


def _torch_device(ctx):
    try:
        import torch
    except ImportError:
        return None
    if ctx.L.chmc_backend() != b"hip:gfx950" or not torch.cuda.is_available():
        return None
    return torch.device("cuda", ctx.device)


def init_objective_and_grad_device(ctx, u_v_dev_ptr, grad_dev_ptr):
    """init_objective of the reference's finder (sde/mici_extensions.py:1706-1737) and its gradient for every chain, on
    device buffers: 1/2 sum_t r_t^2 + T log sigma + 1/2 |u_v|^2 with r_t = (y_t - obs_func(x_t)) / sigma.  For a fixed
    observation noise this is the unconstrained comparator's target (:82-205), which the library evaluates with one
    forward scan and ONE adjoint sweep per chain (chmc_neg_log_dens_and_grad_device): no Jacobian blocks, no Gram
    factors, no grad-log-det.  Returns the values [B] (host); the gradient is written to grad_dev [B, U + NV]."""
    return ctx.neg_log_dens_and_grad_device(u_v_dev_ptr, grad_dev_ptr, use_gaussian_splitting=False)


def _adam_on_device(ctx, rng, adam_step_size, max_iters, max_init_tries, threshold, slow_progress_ratio, check_iter,
                    max_num_tries, log, max_parallel_tries=16, _calls=None):
    """The finder with (u_v, m, v) and the gradient resident in HBM: per Adam iteration TWO library calls -- objective +
    gradient + row statistics (scan, adjoint sweep, one [B, 3] read-back), then the Adam step (one kernel, one [B, 2]
    upload); the [B, Q] arrays never cross PCIe.  Same restart rules as the host loop below.

    The B rows of the context are SLOTS: slot s starts as try 0 of chain s.  The reference's finder (:1741-1789) runs a
    chain's tries one after the other and keeps the first that reaches the threshold; a batch would then iterate for its
    unluckiest chain (boarding-school SIR with sigma = generate_σ_y(u), 1 024 chains: up to 14 tries of 100+ iterations)
    with nearly every row idle.  Rows of finished chains are therefore handed to the chains still searching, which run their
    next tries k+1, k+2, ... side by side; the chain's result is still its FIRST successful try in try order (a later try
    that succeeds earlier waits, frozen, until every earlier one has failed), so each chain's answer is distributed as
    with sequential tries, and `tries` counts as the reference does."""
    import torch
    if _calls is None:                                     # the two library calls of an iteration, on device pointers
        dev = _torch_device(ctx)
        sync = lambda: torch.cuda.synchronize(dev)         # (the library enqueues on its own stream)
        objective = lambda u_v, g: ctx.adam_objective_device(u_v.data_ptr(), g.data_ptr())
        adam_update = lambda u_v, m, v, g, coef, b1, b2, eps: ctx.adam_update_device(
            u_v.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), coef, b1, b2, eps)
    else:                                                  # (CPU test of the slot bookkeeping: stand-ins on CPU tensors)
        dev, sync, objective, adam_update = _calls
    B, Q, T = ctx.B, ctx.Q, ctx.T
    nuv = Q - T
    var_sigma = getattr(ctx, "variable_sigma", False)
    isig = ctx.U - 1                                       # index of log sigma in u (variable observation noise)
    u_v = torch.from_numpy(rng.standard_normal((B, nuv))).to(dev)
    m, v, g = torch.zeros_like(u_v), torch.zeros_like(u_v), torch.empty_like(u_v)
    t_adam = np.zeros(B)
    prev = np.full(B, np.inf)
    it_in_try = np.zeros(B, dtype=np.int64)
    # slots: 0 = iterating, 1 = reached the threshold (frozen), 2 = free
    slot_state = np.zeros(B, dtype=np.int64)
    owner, tryno = np.arange(B), np.zeros(B, dtype=np.int64)
    # chains: the tries handed out so far (try -> slot or -1 = failed), the first try not known to have failed, the winner
    status = [{0: c} for c in range(B)]
    kstar = np.zeros(B, dtype=np.int64)
    next_try = np.ones(B, dtype=np.int64)
    winner = np.full(B, -1, dtype=np.int64)
    result = torch.zeros_like(u_v)                         # every chain's first successful try
    n_running = np.ones(B, dtype=np.int64)
    limit = max_num_tries * max_init_tries
    b1, b2, eps = 0.9, 0.999, 1e-8                         # jax.example_libraries.optimizers.adam defaults
    ls_host = torch.empty(B, dtype=torch.float64)
    if dev.type == "cuda":
        ls_host = ls_host.pin_memory()
    sync()                                                 # (the draws are on the device before the library reads them)

    def release(c, s):
        if slot_state[s] == 0:
            n_running[c] -= 1
        slot_state[s] = 2

    def resolve(c):
        """first try in try order that has not failed: the winner if it has reached the threshold.  Its point moves to the
        result buffer and ALL the chain's slots become free"""
        while status[c].get(int(kstar[c])) == -1:
            kstar[c] += 1
        s = status[c].get(int(kstar[c]))
        if s is not None and slot_state[s] == 1:
            winner[c] = s
            result[c] = u_v[s]
            for k, s2 in status[c].items():
                if s2 >= 0 and slot_state[s2] != 2 and owner[s2] == c and tryno[s2] == k:
                    release(c, s2)

    for _ in range(max_iters * max_num_tries):
        # log sigma: a number, or generate_sigma(u) = exp(u[dim_z]) per slot: B doubles read back per iteration, the copy
        # (torch's stream) runs beside the library's scan and adjoint sweep (its own stream; library calls return synchronised)
        if var_sigma:
            ls_host.copy_(u_v[:, isig], non_blocking=True)
        st = objective(u_v, g)                             # objective, |u_v|^2, gradient finite: one read-back
        val, sq, gfin = st[:, 0], st[:, 1], st[:, 2] != 0.0
        if var_sigma:
            sync()
            log_sigma = ls_host.numpy()
        else:
            log_sigma = np.log(float(ctx.sigma))
        with np.errstate(invalid="ignore", over="ignore"):
            msq = 2.0 * (val - T * log_sigma - 0.5 * sq) / T   # mean squared residual
            running = slot_state == 0
            reached = running & np.isfinite(msq) & (msq < threshold)
            stalled = (it_in_try % check_iter == 0) & (it_in_try > 0) & (it_in_try < max_iters // 2) & (
                msq / prev > slow_progress_ratio)
        failed = running & ~reached & (~np.isfinite(msq) | ~gfin | stalled | (it_in_try >= max_iters))
        upd = (it_in_try % check_iter == 0) & running & ~failed
        prev = np.where(upd, msq, prev)
        touched = set()
        for s in np.flatnonzero(reached):
            slot_state[s] = 1
            n_running[owner[s]] -= 1
            touched.add(int(owner[s]))
        for s in np.flatnonzero(failed):
            c = int(owner[s])
            status[c][int(tryno[s])] = -1
            release(c, s)
            touched.add(c)
        for c in touched:
            if winner[c] < 0:
                resolve(c)
        if (winner >= 0).all():
            break
        # hand the free slots to the chains still searching (slots only change hands when something happened): one more
        # try per chain and turn, fewest running tries first
        fresh = []
        if touched:
            free = np.flatnonzero(slot_state == 2)
            searching = np.flatnonzero(winner < 0)
            while free.size:
                cand = searching[(n_running[searching] < max_parallel_tries) & (next_try[searching] < limit)]
                if cand.size == 0:
                    break
                cand = cand[np.argsort(n_running[cand], kind="stable")][:free.size]
                take, free = free[:cand.size], free[cand.size:]
                owner[take], tryno[take], slot_state[take] = cand, next_try[cand], 0
                for c, s_ in zip(cand.tolist(), take.tolist()):
                    status[c][int(next_try[c])] = s_
                next_try[cand] += 1
                n_running[cand] += 1
                fresh.extend(take.tolist())
            for c in touched:
                if winner[c] < 0 and n_running[c] == 0 and not any(
                        s2 >= 0 and slot_state[s2] == 1 and owner[s2] == c for s2 in status[c].values()):
                    raise RuntimeError(f"Did not find valid state in {max_num_tries} tries.")
        idle = np.flatnonzero((slot_state == 2) & (reached | failed | np.isin(owner, list(touched)))) if touched else []
        if len(idle):                                      # a slot left idle holds a harmless point (a failed try may be NaN)
            idx = torch.from_numpy(np.asarray(idle)).to(dev)
            u_v[idx], m[idx], v[idx], g[idx] = 0.0, 0.0, 0.0, 0.0
        if fresh:
            fr = np.asarray(fresh)
            idx = torch.from_numpy(fr).to(dev)
            u_v[idx] = torch.from_numpy(rng.standard_normal((len(fresh), nuv))).to(dev)
            m[idx], v[idx] = 0.0, 0.0
            g[idx] = 0.0                                   # (a fresh try's moments start from zero at its next gradient)
            t_adam[fr], it_in_try[fr], prev[fr] = 0.0, 0, np.inf
        if fresh or len(idle):
            sync()
        step = slot_state == 0
        if fresh:
            step[np.asarray(fresh)] = False
        t_adam[step] += 1
        # Adam moments in place for every slot (a frozen or free slot's moments are never used again, a fresh try's were
        # zeroed above and its gradient dropped); only the parameter update is masked: one library kernel
        tt = np.maximum(t_adam, 1.0)
        coef = np.stack([1.0 / (1 - b2 ** tt), np.where(step, adam_step_size / (1 - b1 ** tt), 0.0)], 1)
        adam_update(u_v, m, v, g, coef, b1, b2, eps)
        it_in_try[step] += 1
        if log is not None and int(it_in_try.max()) % check_iter == 0:
            log(f"  adam (device): {int((winner >= 0).sum())} of {B} chains below the threshold, "
                f"{int((slot_state == 0).sum())} tries running")
    else:
        raise RuntimeError("Did not find valid states within the iteration budget.")
    tries = kstar + 1
    u_v = result
    if _calls is not None:
        return u_v, tries, status
    # n := residuals puts the point on the manifold (:1767-1775): one state evaluation at [u_v, 0] gives obs_func(x_t) - y_t
    q = np.concatenate([u_v.cpu().numpy(), np.zeros((B, T))], 1)
    xo0 = np.zeros((B, T, ctx.X))
    ctx.set_state(q, None, xo0, 0)
    sigma = np.exp(q[:, isig:isig + 1]) if var_sigma else float(ctx.sigma)
    res = -ctx.constr() / sigma
    assert (np.mean(res ** 2, 1) < threshold * (1 + 1e-9)).all()
    q[:, nuv:] = res
    ctx.set_state(q, None, xo0, 0)
    ctx.update_x_obs_seq()
    xo = ctx.get_state(want_p=False)[2]
    ctx.set_state(q, None, xo, 0)
    return q, xo, tries


def find_initial_states_by_gradient_descent_noisy_system(ctx, rng, adam_step_size=2e-2, max_iters=1000, max_init_tries=100,
                                                         threshold=1.0, slow_progress_ratio=0.8, check_iter=100,
                                                         max_num_tries=10, log=None, device_resident=None):
    """find_initial_state_by_gradient_descent_noisy_system (sde/mici_extensions.py:1679-1801), for every chain of `ctx`
    at once (the SIR script's initialisation, scripts/sir_model_chmc_experiment.py:103-109).

    Adam descent on the negative log posterior density of the noisy-observation model in (u, v_0, v_seq),
        1/2 sum_t r_t^2 + T log sigma + 1/2 |u_v|^2,   r_t = (y_t - obs_func(x_t(u_v))) / sigma
    (sigma fixed, or sigma = generate_σ_y(u) = exp(u[dim_z]) for a context with variable observation noise),
    until the mean squared residual drops below `threshold`; the point is then put on the manifold by setting the
    observation-noise components to the residuals (:1767-1775).  The residuals and their gradient come from the
    library: `ctx` must hold the whole observation sequence in ONE sub-sequence (num_obs_per_subseq >= num_obs, i.e.
    K = 1, the SIR configuration), where the constraint function at n = 0 is obs_func(x_t) - y_t for the full scan and
    J^T lambda is its exact adjoint.  Chains restart from a fresh draw when Adam diverges or stalls, as in the reference.
    Leaves the found states set on `ctx` (zero momentum) and returns (q [B, Q], x_obs_seq [B, T, X], tries [B]).

    device_resident (None: when possible): the iteration runs with (u_v, m, v) resident in HBM and the objective / gradient
    from the library's scan + single adjoint sweep (`_adam_on_device`), for a fixed observation noise and for
    sigma = generate_σ_y(u) alike (round 4; the gradient's u[dim_z] component is T - sum r^2 + u[dim_z]); the host loop below
    (full state evaluation per iteration through the per-operator entry points) is the independent second implementation."""
    if not ctx.noisy or ctx.num_blocks != 1 or ctx.num_partition != 1:
        raise ValueError("needs a noisy-observation context with a single sub-sequence (num_obs_per_subseq >= num_obs)")
    B, Q, T = ctx.B, ctx.Q, ctx.T
    nuv = Q - T
    var_sigma = getattr(ctx, "variable_sigma", False)
    if device_resident is not False and _torch_device(ctx) is not None:
        return _adam_on_device(ctx, rng, adam_step_size, max_iters, max_init_tries, threshold, slow_progress_ratio, check_iter,
                               max_num_tries, log)
    if device_resident is True:
        raise ValueError("the device-resident finder needs a CUDA-capable torch")
    iσ = ctx.U - 1                                         # index of log sigma in u (variable observation noise)
    xo0 = np.zeros((B, T, ctx.X))

    def residuals_and_grad(u_v):
        q = np.concatenate([u_v, np.zeros((B, T))], 1)
        ctx.set_state(q, None, xo0, 0)
        c = ctx.constr()                                   # obs_func(x_t) - y_t  (n = 0: independent of sigma)
        sigma = np.exp(u_v[:, iσ:iσ + 1]) if var_sigma else float(ctx.sigma)
        g = ctx.rmult_by_jacob_constr(c / sigma ** 2)[:, :nuv] + u_v
        if var_sigma:                                      # d/du_sigma [1/2 sum c^2 / sigma^2 + T log sigma]
            g[:, iσ] += T - np.sum((c / sigma) ** 2, 1)
        return -c / sigma, g

    u_v = rng.standard_normal((B, nuv))
    m, v = np.zeros_like(u_v), np.zeros_like(u_v)
    t_adam = np.zeros(B)
    done = np.zeros(B, dtype=bool)
    tries = np.ones(B, dtype=np.int64)
    prev = np.full(B, np.inf)
    it_in_try = np.zeros(B, dtype=np.int64)
    res_found = np.zeros((B, T))
    b1, b2, eps = 0.9, 0.999, 1e-8                         # jax.example_libraries.optimizers.adam defaults
    for _ in range(max_iters * max_num_tries):
        r, g = residuals_and_grad(u_v)
        msq = np.mean(r ** 2, 1)
        newly = ~done & np.isfinite(msq) & (msq < threshold)
        res_found[newly] = r[newly]
        done |= newly
        if done.all():
            break
        stalled = (it_in_try % check_iter == 0) & (it_in_try > 0) & (it_in_try < max_iters // 2) & (
            msq / prev > slow_progress_ratio)
        restart = ~done & (~np.isfinite(msq) | ~np.isfinite(g).all(1) | stalled | (it_in_try >= max_iters))
        upd = (it_in_try % check_iter == 0) & ~restart
        prev = np.where(upd, msq, prev)
        if restart.any():
            if (tries[restart] >= max_num_tries * max_init_tries).any():
                raise RuntimeError(f"Did not find valid state in {max_num_tries} tries.")
            n = int(restart.sum())
            u_v[restart] = rng.standard_normal((n, nuv))
            m[restart], v[restart], t_adam[restart], it_in_try[restart], prev[restart] = 0.0, 0.0, 0.0, 0, np.inf
            tries[restart] += 1
        step = ~done & ~restart
        t_adam[step] += 1
        m[step] = b1 * m[step] + (1 - b1) * g[step]
        v[step] = b2 * v[step] + (1 - b2) * g[step] ** 2
        mh = m[step] / (1 - b1 ** t_adam[step])[:, None]
        vh = v[step] / (1 - b2 ** t_adam[step])[:, None]
        u_v[step] -= adam_step_size * mh / (np.sqrt(vh) + eps)
        it_in_try[step] += 1
        if log is not None and int(it_in_try.max()) % check_iter == 0:
            log(f"  adam: {int(done.sum())} of {B} chains below the threshold, median mean r^2 {np.median(msq):.3g}")
    else:
        raise RuntimeError("Did not find valid states within the iteration budget.")
    q = np.concatenate([u_v, res_found], 1)                # n := residuals puts the point on the manifold
    ctx.set_state(q, None, xo0, 0)
    ctx.update_x_obs_seq()
    xo = ctx.get_state(want_p=False)[2]
    ctx.set_state(q, None, xo, 0)
    return q, xo, tries
