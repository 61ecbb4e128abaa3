"""The minimal caller (sampling.py): snapshot / restore semantics and detailed-balance sanity on a tiny problem,
through the TEST-ONLY emulation build (CPU)."""
import os
import numpy as np
from test_emu_logic import emu_lib  # noqa: F401
from helpers import make_case, make_ctx


def test_snapshot_restore_and_head(emu_lib):  # noqa: F811
    case = make_case("fhn", 6, 4, 2, True, B=3, seed=41)
    ctx = make_ctx(case)
    qq = np.repeat(case["q"][:1], 3, 0)
    xx = np.repeat(case["x_obs"][:1], 3, 0)
    ctx.set_state(qq, None, xx, 0)
    ctx.sample_momentum(1, 1, 0)
    q0, p0, _, _ = ctx.get_state()
    ld0, g0 = ctx.log_det_sqrt_gram(), ctx.grad_log_det_sqrt_gram()
    ctx.snapshot()
    r = ctx.leapfrog_step(np.array([0.05, -0.05, 0.05]), constraint_tol=1e-9, position_tol=1e-8)
    assert (r["status"] == 0).all()
    q1, p1, _, _ = ctx.get_state()
    ctx.restore(np.array([1, 0, 1]))
    q2, p2, _, _ = ctx.get_state()
    assert np.array_equal(q2[0], q0[0]) and np.array_equal(p2[2], p0[2]) and np.array_equal(q2[1], q1[1])
    ld2, g2 = ctx.log_det_sqrt_gram(), ctx.grad_log_det_sqrt_gram()
    np.testing.assert_allclose(ld2[[0, 2]], ld0[[0, 2]], rtol=1e-13)
    np.testing.assert_allclose(g2[0], g0[0], rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(ctx.get_head(6), q2[:, :6])
    # a step from the restored state reproduces the first step
    r = ctx.leapfrog_step(np.array([0.05, -0.05, 0.05]), active=np.array([1, 0, 1]), constraint_tol=1e-9, position_tol=1e-8)
    q3, _, _, _ = ctx.get_state()
    np.testing.assert_allclose(q3[0], q1[0], rtol=0, atol=1e-12)
    ctx.close()


def test_static_sampler_runs_and_adapts(emu_lib):  # noqa: F811
    from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc
    from manifold_mcmc_for_diffusions_amd import example_models as em
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    y = em.simulate_fhn_observations(6, 0.2, 50, seed=5, sigma=0.1)
    ctx = ChmcContext("fhn", 0.2, 4, 2, y[:, 0], sigma=0.1, num_chains=4)
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, 4, y, 4, True, seed=7)
    ctx.set_state(q, None, xo, 0)
    res = sample_static_chmc(ctx, 12, 3, 0.05, seed=3, n_adapt=6)
    assert res["heads"].shape == (12, 4, 6) and np.isfinite(res["heads"]).all()
    assert 0.0 <= res["accept_stat"].min() and res["accept_stat"].max() <= 1.0
    assert np.abs(ctx.constr()).max() < 1e-8  # every retained state lies on the manifold
    assert res["final_step_size"] > 0
    ctx.close()


def test_trace_files_and_summary(emu_lib, tmp_path):  # noqa: F811
    """trace_dir output: memory-mapped [chain, draw, ...] arrays per traced variable and a summary.json with the
    reference's run totals (scripts/utils.py:368-381)."""
    import json
    from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc
    from manifold_mcmc_for_diffusions_amd import example_models as em
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    y = em.simulate_fhn_observations(6, 0.2, 50, seed=5, sigma=0.1)
    ctx = ChmcContext("fhn", 0.2, 4, 2, y[:, 0], sigma=0.1, num_chains=3)
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, 4, y, 3, True, seed=7)
    ctx.set_state(q, None, xo, 0)

    def trace_func(head, ham):  # scripts/fhn_model_noisy_obs_chmc_experiment.py:82-99
        z = np.stack([em.fhn.generate_z(u) for u in head[:, :4]])
        return {"σ": z[:, 0], "ϵ": z[:, 1], "γ": z[:, 2], "β": z[:, 3], "hamiltonian": ham,
                "x_0": np.stack([em.fhn.generate_x_0(zz, v) for zz, v in zip(z, head[:, 4:6])])}

    d = str(tmp_path / "out")
    res = sample_static_chmc(ctx, 10, 2, 0.05, seed=3, n_adapt=4, trace_dir=d, trace_func=trace_func)
    x0 = np.load(os.path.join(d, "trace_x_0.npy"), mmap_mode="r")
    sig = np.load(os.path.join(d, "trace_σ.npy"))
    assert x0.shape == (3, 10, 2) and sig.shape == (3, 10)
    np.testing.assert_array_equal(x0[:, -1], trace_func(res["heads"][-1], np.zeros(3))["x_0"])
    s = json.load(open(os.path.join(d, "summary.json")))
    assert {"mean", "sd", "r_hat", "ess_bulk", "total_sampling_time", "final_integrator_step_size",
            "total_leapfrog_step_calls", "total_constr_calls"} <= set(s)
    assert s["total_leapfrog_step_calls"] == 10 * 2 and "x_0[1]" in s["mean"] and "σ" in s["sd"]
    assert s["final_integrator_step_size"] == res["final_step_size"]
    ctx.close()


def test_rhat_and_ess_on_known_processes():
    from manifold_mcmc_for_diffusions_amd.traces import split_rhat_and_ess, summarize
    rng = np.random.default_rng(0)
    iid = rng.standard_normal((8, 2000))
    r, e = split_rhat_and_ess(iid)
    assert abs(r - 1.0) < 0.01 and 0.8 * 16000 < e < 1.25 * 16000
    ar = np.zeros((8, 4000))  # AR(1), rho = 0.9: ESS = N (1 - rho) / (1 + rho)
    eps = rng.standard_normal(ar.shape)
    for t in range(1, ar.shape[1]):
        ar[:, t] = 0.9 * ar[:, t - 1] + eps[:, t]
    r, e = split_rhat_and_ess(ar)
    assert abs(r - 1.0) < 0.05 and 0.6 < e / (32000 * 0.1 / 1.9) < 1.6
    shifted = iid + np.arange(8)[:, None]  # chains that disagree
    assert split_rhat_and_ess(shifted)[0] > 1.5
    s = summarize({"a": iid, "b": np.stack([iid, ar[:, :2000]], -1)})
    assert set(s["mean"]) == {"a", "b[0]", "b[1]"}


def test_jittered_lengths_and_warmup_backoff(emu_lib):  # noqa: F811
    """jitter_length draws every chain's number of steps per transition; chains that cannot move back their step
    size off during warm-up only.  Every retained state stays on the manifold and the run is reproducible."""
    from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc
    from manifold_mcmc_for_diffusions_amd import example_models as em
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    y = em.simulate_fhn_observations(6, 0.2, 50, seed=5, sigma=0.1)
    outs = []
    for _ in range(2):
        ctx = ChmcContext("fhn", 0.2, 4, 2, y[:, 0], sigma=0.1, num_chains=4)
        q, xo, _ = fhn_initial_states(em.fhn, 0.2, 4, y, 4, True, seed=7)
        ctx.set_state(q, None, xo, 0)
        calls0 = ctx.counters()["leapfrog_step"]
        res = sample_static_chmc(ctx, 10, 6, 0.05, seed=3, n_adapt=5, jitter_length=True)
        assert ctx.counters()["leapfrog_step"] - calls0 <= 10 * 6  # chains stop after their own length
        assert np.abs(ctx.constr()).max() < 1e-8 and np.isfinite(res["heads"]).all()
        outs.append(res["heads"])
        ctx.close()
    np.testing.assert_array_equal(outs[0], outs[1])


def test_block_diagonal_metric_adapter_statistics():
    """OnlineBlockDiagonalMetricAdapter (sde/mici_extensions.py:1804-1931): per-chain Welford statistics combined over
    chains equal the pooled sample covariance; regularisation towards reg_scale * I; metric = blockdiag(cov^-1, I)."""
    import pytest
    from manifold_mcmc_for_diffusions_amd.adapters import OnlineBlockDiagonalMetricAdapter
    from manifold_mcmc_for_diffusions_amd.errors import AdaptationError
    rng = np.random.default_rng(4)
    B, n, d, Q = 5, 40, 4, 9
    A = rng.standard_normal((d, d))
    draws = rng.standard_normal((n, B, Q))
    draws[..., :d] = draws[..., :d] @ A.T + rng.standard_normal(d) * 3.0
    ad = OnlineBlockDiagonalMetricAdapter(d, reg_iter_offset=5, reg_scale=1e-3)
    st = ad.initialize(draws[0])
    for t in range(n):
        ad.update(st, draws[t])
    for c in range(B):  # Welford per chain (:1868-1879)
        np.testing.assert_allclose(st["mean"][c], draws[:, c, :d].mean(0), atol=1e-12)
        np.testing.assert_allclose(st["sum_diff_outer"][c], np.cov(draws[:, c, :d].T) * (n - 1), atol=1e-10)
    metric = ad.finalize(st)
    pooled = np.cov(draws[..., :d].reshape(-1, d).T)  # Schubert & Gertz combination = pooled covariance (:1899-1918)
    N = n * B
    expect = pooled * N / (5 + N) + 1e-3 * 5 / (5 + N) * np.eye(d)  # _regularize_covar_est (:1881-1890)
    np.testing.assert_allclose(np.linalg.inv(metric.blocks[0].array), expect, rtol=1e-9, atol=1e-12)
    assert metric.blocks[1].size == Q - d
    v = rng.standard_normal((3, Q))
    np.testing.assert_allclose((metric.inv @ v)[:, :d], v[:, :d] @ expect.T, rtol=1e-9)
    np.testing.assert_allclose((metric.inv @ v)[:, d:], v[:, d:])
    np.testing.assert_allclose(metric @ (metric.inv @ v), v, rtol=1e-9, atol=1e-12)
    s = metric.sqrt @ v
    np.testing.assert_allclose(s[:, :d], v[:, :d] @ np.linalg.cholesky(metric.blocks[0].array).T, rtol=1e-10)
    # a mask keeps chains out of a draw; a single draw is not enough (:1919-1923)
    st2 = ad.initialize(draws[0])
    ad.update(st2, draws[0], mask=np.array([1, 0, 0, 0, 0]))
    assert st2["iter"].tolist() == [1, 0, 0, 0, 0]
    with pytest.raises(AdaptationError):
        ad.finalize(st2)


def test_static_sampler_with_metric_adaptation(emu_lib):  # noqa: F811
    from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc
    from manifold_mcmc_for_diffusions_amd.adapters import OnlineBlockDiagonalMetricAdapter
    from manifold_mcmc_for_diffusions_amd import example_models as em
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    y = em.simulate_fhn_observations(6, 0.2, 50, seed=5, sigma=0.1)
    ctx = ChmcContext("fhn", 0.2, 4, 2, y[:, 0], sigma=0.1, num_chains=4)
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, 4, y, 4, True, seed=7)
    ctx.set_state(q, None, xo, 0)
    ad = OnlineBlockDiagonalMetricAdapter(4)
    res = sample_static_chmc(ctx, 14, 3, 0.05, seed=3, n_adapt=8, metric_adapter=ad)
    # the installed M_0 is the adapter's result on draws 2..5 of the global parameters (warm-up fractions 0.25 .. 0.75)
    st = ad.initialize(q)
    for t in range(2, 6):
        ad.update(st, res["heads"][t])
    np.testing.assert_allclose(res["metric_M_0"], ad.finalize(st).blocks[0].array, rtol=1e-10)
    np.testing.assert_allclose(ctx.M_0, res["metric_M_0"])
    assert np.isfinite(res["heads"]).all() and np.abs(ctx.constr()).max() < 1e-8
    # tangent momenta with respect to the adapted metric: J M^-1 p = 0
    ctx.sample_momentum(3, 99)
    _, p, _, _ = ctx.get_state()
    p[:, :4] = p[:, :4] @ np.linalg.inv(ctx.M_0).T
    assert np.abs(ctx.lmult_by_jacob_constr(p)).max() < 1e-8 * np.abs(p).max()
    ctx.close()


def test_adam_finder_parallel_tries_keep_the_first_successful_try_in_try_order():
    """Slot bookkeeping of the device-resident initial-state finder (init._adam_on_device) with stand-ins for its two library
    calls: rows of finished chains run the later tries of the chains still searching side by side, but a chain's result must
    be its FIRST successful try in try order, as the reference's sequential tries give (sde/mici_extensions.py:1741-1789).
    Landscape: a try whose draw has u[0] > 0 never gets below the threshold (stalls, or is NaN at once for u[0] > 1.5)."""
    import torch
    from types import SimpleNamespace
    from manifold_mcmc_for_diffusions_amd import init
    B, T, nuv = 48, 6, 12
    ctx = SimpleNamespace(B=B, Q=nuv + T, T=T, U=3, sigma=1.0, variable_sigma=False)
    n_calls = [0]

    def objective(u_v, g):
        n_calls[0] += 1
        u = u_v.numpy()
        bad = u[:, 0] > 0
        h = 10.0 * np.mean(u[:, 1:] ** 2, 1) + np.where(bad, 2.0, 0.0)
        val = 0.5 * T * h + 0.5 * np.sum(u ** 2, 1)
        val = np.where(u[:, 0] > 1.5, np.nan, val)
        gr = u.copy()
        gr[:, 0] = 0.0  # (the draw's fate is fixed: u[0] never moves)
        gr[:, 1:] += 0.5 * T * 20.0 * u[:, 1:] / (nuv - 1)
        g.copy_(torch.from_numpy(gr))
        return np.stack([val, np.sum(u ** 2, 1), np.ones(B)], 1)

    def adam_update(u_v, m, v, g, coef, b1, b2, eps):
        c = torch.from_numpy(coef)
        m.mul_(b1).add_((1 - b1) * g)
        v.mul_(b2).add_((1 - b2) * g * g)
        u_v.sub_(c[:, 1:2] * m / (torch.sqrt(v * c[:, 0:1]) + eps))

    class Rng:  # records the draws in order
        def __init__(self):
            self.r, self.rows = np.random.default_rng(5), []
        def standard_normal(self, shape):
            x = self.r.standard_normal(shape)
            self.rows.append(x.copy())
            return x

    rng = Rng()
    u_v, tries, status = init._adam_on_device(ctx, rng, 0.1, 1000, 100, 1.0, 0.8, 100, 10, None, max_parallel_tries=4,
                                              _calls=(torch.device("cpu"), lambda: None, objective, adam_update))
    u = u_v.numpy()
    first = rng.rows[0]
    assert (u[:, 0] <= 0).all() and (10.0 * np.mean(u[:, 1:] ** 2, 1) < 1.0).all()
    ok0 = first[:, 0] <= 0
    assert (tries[ok0] == 1).all() and (tries[~ok0] >= 2).all() and tries.max() >= 3
    # a chain that kept its first try returns the descendant of ITS draw
    assert (u[ok0, 0] == first[ok0, 0]).all()
    for c in range(B):
        assert all(status[c][k] == -1 for k in range(int(tries[c]) - 1)) and status[c][int(tries[c]) - 1] >= 0
    # the later tries ran side by side: far fewer iterations than the sum over the unluckiest chain's tries
    # (a stalling try is given up at its second check, after 200 iterations: sequential tries would need 200 per failed try of
    # the unluckiest chain, about 1 000 iterations here)
    assert tries.max() >= 5 and n_calls[0] < 500, (n_calls[0], tries.max())
