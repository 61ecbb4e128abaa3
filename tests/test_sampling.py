"""The minimal caller (sampling.py): snapshot / restore semantics and detailed-balance sanity on a tiny problem,
through the TEST-ONLY emulation build (CPU)."""
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
