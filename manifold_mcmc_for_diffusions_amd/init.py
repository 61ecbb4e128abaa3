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
                                               generate_x_obs_seq_init, noisy, u=None, v_0=None):
    """Returns (q, x_obs_seq) for one chain; RNG draw order follows the reference (:1528-1532)."""
    δ = obs_interval / num_steps_per_obs
    u = rng.standard_normal(model.dim_z) if u is None else np.asarray(u, dtype=np.float64)
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
