"""Pins between the independent statements of the mathematics (all CPU):
  * one-step maps: hand-written torch closed forms (oracle/py/models.py) vs the C code
    generated symbolically from drift / diffusion coefficients (tools/gen_models.py) vs the NumPy host models;
  * C oracle (analytic adjoint / tangent sweeps) vs the torch.func autodiff oracle, operator by operator;
  * C oracle derivatives vs central finite differences."""
import numpy as np
import pytest
import torch
from oracle import c_oracle
from oracle.py import models as omodels, system as osys
from manifold_mcmc_for_diffusions_amd import example_models as em
from helpers import make_case, random_q


@pytest.mark.parametrize("model", ["fhn", "sir", "fhn_nb"])
def test_one_step_maps_agree(model):
    rng = np.random.default_rng(0)
    m_t, m_n = omodels.MODELS[model], em.MODELS[model]
    dl = 0.013
    for _ in range(20):
        q = random_q(model, 1, 1, False, 1, rng, v_scale=1.0, u_scale=0.5)[0]
        u, v0, v = q[:4], q[4:4 + m_n.dim_v_0], q[4 + m_n.dim_v_0:]
        z_t = m_t.generate_z(osys.T(u))
        x0_t = m_t.generate_x_0(z_t, osys.T(v0))
        x1_t = m_t.forward_func(z_t, x0_t, osys.T(v), dl).numpy()
        z_n = m_n.generate_z(u)
        np.testing.assert_allclose(z_n, z_t.numpy(), rtol=1e-14)
        x0_n = m_n.generate_x_0(z_n, v0)
        np.testing.assert_allclose(x0_n, x0_t.numpy(), rtol=1e-14)
        np.testing.assert_allclose(m_n.forward_func(z_n, x0_n, v, dl), x1_t, rtol=1e-12, atol=1e-13)
        osy = c_oracle.OracleSystem(model, dl, 1, None, np.zeros(1), sigma=None)  # T = S = 1: x_obs = f(z, x_0, v)
        np.testing.assert_allclose(osy.generate_x_obs_seq(q)[0], x1_t, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(m_n.obs_func(x1_t), m_t.obs_func(osys.T(x1_t)).numpy(), rtol=1e-14)


def test_sir_clipping_matches_reference_semantics():
    """sir.py:54-70: components at / below -500 are frozen at -500, the others step normally."""
    z = np.array([0.4, 0.3, 1.0, 0.05])
    x = np.array([-600.0, 2.0, 0.5])
    v = np.array([0.3, -0.2, 0.1])
    a = em.sir.forward_func(z, x, v, 0.01)
    b = omodels.sir.forward_func(osys.T(z), osys.T(x), osys.T(v), 0.01).numpy()
    np.testing.assert_allclose(a, b, rtol=1e-13)
    assert a[0] == -500.0 and np.isfinite(a).all()


def rowslot(osy, jac):
    du = torch.cat([b.reshape(-1, b.shape[-1]) for b in jac[0]]).numpy()
    dv = np.zeros((osy.rmax, osy.NV))
    col = 0
    for b in jac[1]:
        b = b.numpy()
        if b.ndim == 2:
            b = b[None]
        for mm_ in range(b.shape[0]):
            r, nc = b[mm_].shape
            dv[:r, col:col + nc] = b[mm_]
            col += nc
    return du, dv


CASES = [("fhn", 6, 4, 2, True, False), ("fhn", 7, 5, 3, False, True), ("fhn", 9, 3, 4, True, False),
         ("fhn", 5, 4, None, True, False), ("sir", 5, 6, None, True, False), ("sir", 6, 8, 2, True, False),
         # the notebook's model: the C side is generated code, the autodiff side the hand-written oracle/py/models.py
         ("fhn_nb", 7, 5, 3, False, True), ("fhn_nb", 6, 4, 2, True, False)]


# variable observation noise (generate_σ callable, dim_u = dim_z + 1: scripts/sir_model_chmc_experiment.py:44,58,77)
VS_CASES = [("sir", 5, 6, None, True, False), ("sir", 6, 8, 2, True, False), ("fhn", 6, 4, 2, True, False),
            ("fhn", 7, 5, 3, True, True)]


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,var_sigma",
                         [c + (False,) for c in CASES] + [c + (True,) for c in VS_CASES])
def test_c_oracle_vs_autodiff_oracle(model, T, S, R, noisy, gaussian, var_sigma):
    case = make_case(model, T, S, R, noisy, B=2, seed=21, gaussian=gaussian, var_sigma=var_sigma)
    osy = case["osys"]
    ref = osys.make_system(omodels.MODELS[model], case["obs_interval"], S, R, case["y"][:, None], sigma=case["sigma"],
                           use_gaussian_splitting=gaussian)
    q, xo = case["q"][1], case["x_obs"][1]  # chain 1 is off the manifold: non-trivial constraint values
    rng = case["rng"]
    for part in range(osy.num_partition):
        st = osys.ConditionedDiffusionHamiltonianState(q, xo, part)
        jac, chol = ref.jacob_constr_blocks(st), ref.chol_gram_blocks(st)
        du_r, dv_r = rowslot(osy, jac)
        c, du, dv = osy.jacob_constr_blocks(q, xo, part)
        cC, cD, ld, grad = osy.gram_ops(q, xo, part)
        np.testing.assert_allclose(c, ref.constr(st), atol=1e-12)
        np.testing.assert_allclose(du, du_r, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(dv, dv_r, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cC, chol[0].numpy(), rtol=1e-9, atol=1e-11)
        assert abs(ld - ref.log_det_sqrt_gram(st)) < 1e-10 * max(1.0, abs(ld))
        g_ref = ref.grad_log_det_sqrt_gram(st)
        np.testing.assert_allclose(grad, g_ref, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(g_ref).max()))
        w, lam = rng.standard_normal(osy.Q), rng.standard_normal(osy.dim_c(part))
        Jw, JTl, Gil, nsc = osy.jacob_products(q, xo, part, w, lam)
        np.testing.assert_allclose(Jw, ref._lmult_by_jacob_constr(*jac, osys.T(w)).numpy(), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(JTl, ref._rmult_by_jacob_constr(*jac, osys.T(lam)).numpy(), rtol=1e-10, atol=1e-11)
        ref_gil = ref._lmult_by_inv_gram(*jac, *chol, osys.T(lam)).numpy()
        np.testing.assert_allclose(Gil, ref_gil, rtol=1e-8, atol=1e-9 * np.abs(ref_gil).max())
        np.testing.assert_allclose(nsc, ref.normal_space_component(st, w), rtol=1e-8, atol=1e-9)
    # dense cross-check of the Woodbury solve and the log-determinant
    Jd = torch.func.jacrev(lambda qq: ref._constr(qq, osys.T(xo), 0))(osys.T(q)).numpy()
    G = Jd @ Jd.T
    lam = rng.standard_normal(G.shape[0])
    _, _, Gil, _ = osy.jacob_products(q, xo, 0, np.zeros(osy.Q), lam)
    np.testing.assert_allclose(Gil, np.linalg.solve(G, lam), rtol=1e-7, atol=1e-9 * np.abs(Gil).max())
    assert abs(osy.gram_ops(q, xo, 0, want_grad=False)[2] - 0.5 * np.linalg.slogdet(G)[1]) < 1e-9


def test_gradient_against_finite_differences():
    case = make_case("fhn", 6, 4, 2, True, B=2, seed=22)
    osy, q, xo = case["osys"], case["q"][1], case["x_obs"][1]
    _, _, _, grad = osy.gram_ops(q, xo, 0)
    rng = np.random.default_rng(1)
    for _ in range(6):
        d = rng.standard_normal(q.shape)
        d /= np.linalg.norm(d)
        eps = 1e-5
        fd = (osy.gram_ops(q + eps * d, xo, 0, want_grad=False)[2] - osy.gram_ops(q - eps * d, xo, 0, want_grad=False)[2]) / (2 * eps)
        assert abs(fd - grad @ d) < 1e-6 * max(1.0, abs(fd))
    c0, du, dv = osy.jacob_constr_blocks(q, xo, 0)
    d = rng.standard_normal(q.shape)
    fdJ = (osy.constr(q + 1e-6 * d, xo, 0) - osy.constr(q - 1e-6 * d, xo, 0)) / 2e-6
    Jw, _, _, _ = osy.jacob_products(q, xo, 0, d, np.zeros(osy.dim_c(0)))
    np.testing.assert_allclose(Jw, fdJ, rtol=1e-6, atol=1e-7)


def test_block_shape_tables_of_the_baseline_configs():
    """SURVEY.md Appendix A (computed from sde/mici_extensions.py:317-351)."""
    y = np.zeros(100)
    o = c_oracle.OracleSystem("fhn", 0.2, 400, 5, y, sigma=0.1)
    assert o.Q == 80106 and (o.dim_c(0), o.dim_c(1)) == (138, 140) and (o.num_blocks(0), o.num_blocks(1)) == (20, 21)
    b0 = [o.block_info(0, b) for b in range(20)]
    assert (b0[0]["nrows"], b0[0]["ncols"]) == (7, 4002) and (b0[1]["nrows"], b0[1]["ncols"]) == (7, 4000)
    assert (b0[-1]["nrows"], b0[-1]["ncols"]) == (5, 4000)
    b1 = [o.block_info(1, b) for b in range(21)]
    assert (b1[0]["nrows"], b1[0]["ncols"]) == (4, 1602) and (b1[-1]["nrows"], b1[-1]["ncols"]) == (3, 2400)
    o = c_oracle.OracleSystem("fhn", 0.2, 400, 5, y, sigma=None)
    assert o.Q == 80006 and (o.dim_c(0), o.dim_c(1)) == (119, 120)
    o = c_oracle.OracleSystem("sir", 1.0, 200, 14, np.zeros(14), sigma=1.0)
    assert o.Q == 8419 and o.num_partition == 1 and o.dim_c(0) == 14 and o.block_info(0, 0)["ncols"] == 8401
    o = c_oracle.OracleSystem("fhn", 0.2, 50, 5, y, sigma=0.1)
    assert o.Q == 10106


def test_linear_interpolation_initial_state():
    """sde/mici_extensions.py:1479-1547: the generated state satisfies the constraint in every partition, and the
    host (NumPy) initialiser agrees with the restatement driven by autodiff."""
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states, find_initial_state_by_linear_interpolation
    y = em.simulate_fhn_observations(10, 0.2, 100, seed=3, sigma=0.1)
    q, xo, rngs = fhn_initial_states(em.fhn, 0.2, 8, y, 3, True, seed=5)
    osy = c_oracle.OracleSystem("fhn", 0.2, 8, 5, y[:, 0], sigma=0.1)
    for c in range(3):
        for part in range(2):
            assert np.abs(osy.constr(q[c], xo[c], part)).max() < 1e-9
        np.testing.assert_allclose(osy.generate_x_obs_seq(q[c]), xo[c], atol=1e-10)
    # sharding invariance of the per-chain generators
    q2, xo2, _ = fhn_initial_states(em.fhn, 0.2, 8, y, 2, True, seed=5, chain_offset=1, total_chains=3)
    np.testing.assert_array_equal(q2, q[1:])
    ref = osys.make_system(omodels.fhn, 0.2, 8, 5, y, sigma=0.1)
    gen = lambda r: np.concatenate((y, r.standard_normal(y.shape) * 0.5), -1)  # noqa: E731
    rs = osys.find_initial_state_by_linear_interpolation(ref, np.random.default_rng(9), gen)
    qh, xoh = find_initial_state_by_linear_interpolation(em.fhn, 0.2, 8, y, np.random.default_rng(9), gen, True)
    np.testing.assert_allclose(qh, rs.pos, atol=1e-11)
    np.testing.assert_allclose(xoh, rs.x_obs_seq, atol=0)


def _spd(rng, n):
    a = rng.standard_normal((n, n))
    return a @ a.T / n + 0.5 * np.eye(n)


@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 6, 4, 2, True), ("fhn", 7, 5, 3, False), ("sir", 5, 6, None, True)])
def test_c_oracle_block_metric_against_dense_algebra(model, T, S, R, noisy):
    """metric = blockdiag(M_0, I) (sde/mici_extensions.py:303-315, 794-798, 1033-1041, 1105-1113, 1202-1259): the oracle's
    Woodbury pieces against dense linear algebra on the autodiff Jacobian."""
    case = make_case(model, T, S, R, noisy, B=2, seed=31)
    osy, rng = case["osys"], case["rng"]
    ref = osys.make_system(omodels.MODELS[model], case["obs_interval"], S, R, case["y"][:, None], sigma=case["sigma"])
    q, xo = case["q"][1], case["x_obs"][1]
    M0 = _spd(rng, 4)
    osy.set_metric(M0)
    Minv = np.eye(osy.Q)
    Minv[:4, :4] = np.linalg.inv(M0)
    for part in range(osy.num_partition):
        Jd = torch.func.jacrev(lambda qq: ref._constr(qq, osys.T(xo), part))(osys.T(q)).numpy()
        G = Jd @ Minv @ Jd.T
        w, lam = rng.standard_normal(osy.Q), rng.standard_normal(G.shape[0])
        cC, _, ld, grad = osy.gram_ops(q, xo, part)
        assert abs(ld - 0.5 * np.linalg.slogdet(G)[1]) < 1e-9
        _, _, Gil, nsc = osy.jacob_products(q, xo, part, w, lam)
        np.testing.assert_allclose(Gil, np.linalg.solve(G, lam), rtol=1e-7, atol=1e-9 * np.abs(Gil).max())
        np.testing.assert_allclose(nsc, Jd.T @ np.linalg.solve(G, Jd @ Minv @ w), rtol=1e-7, atol=1e-9)
        for _ in range(3):  # gradient of the log-determinant with the metric in the Gram matrix
            d = rng.standard_normal(q.shape)
            d /= np.linalg.norm(d)
            fd = (osy.gram_ops(q + 1e-5 * d, xo, part, want_grad=False)[2]
                  - osy.gram_ops(q - 1e-5 * d, xo, part, want_grad=False)[2]) / 2e-5
            assert abs(fd - grad @ d) < 1e-6 * max(1.0, abs(fd))
    # retraction: q' on the manifold, M (q_start - q') in the row space of the previous Jacobian, mu = M dq / dt
    q0, x0 = case["q"][0], case["x_obs"][0]
    Jp = torch.func.jacrev(lambda qq: ref._constr(qq, osys.T(x0), 0))(osys.T(q0)).numpy()
    M = np.linalg.inv(Minv)
    for newton in (True, False):
        qs = q0 + 0.05 * Minv @ rng.standard_normal(osy.Q)
        st, q1, mu, it, ndq, err = osy.project(newton, q0, qs, x0, 0, 0.05)
        assert st == 0 and np.abs(osy.constr(q1, x0, 0)).max() < 1e-9
        r = M @ (qs - q1)
        np.testing.assert_allclose(mu * 0.05, r, rtol=1e-9, atol=1e-12)
        coef = np.linalg.lstsq(Jp.T, r, rcond=None)[0]
        assert np.abs(Jp.T @ coef - r).max() < 1e-9 * max(1.0, np.abs(r).max())
    # one leapfrog step: tangent momentum, conserved constraint, reversibility, energy error O(dt^2)
    ch = c_oracle.OracleChain(osy)
    p = rng.standard_normal(osy.Q)
    p[:4] = np.linalg.cholesky(M0) @ p[:4]  # metric.sqrt @ n (:1257)
    ch.set(q0, p, x0, 0)
    ch.project_mom()
    _, p0, _, _ = ch.get()
    assert np.abs(Jp @ Minv @ p0).max() < 1e-9 * np.abs(p0).max()
    h0 = ch.hamiltonian()
    st, itf, itb, rev = ch.step(0.02)
    q1, p1, _, _ = ch.get()
    assert st == 0 and np.abs(osy.constr(q1, x0, 0)).max() < 1e-9 and rev < 2e-8
    J1 = torch.func.jacrev(lambda qq: ref._constr(qq, osys.T(x0), 0))(osys.T(q1)).numpy()
    assert np.abs(J1 @ Minv @ p1).max() < 1e-8 * np.abs(p1).max()
    assert abs(ch.hamiltonian() - h0) < 0.05 * max(1.0, abs(h0)) * 0.02
    ch.set(q1, -p1, x0, 0)
    st, _, _, _ = ch.step(0.02)
    q2, p2, _, _ = ch.get()
    assert st == 0 and np.abs(q2 - q0).max() < 1e-7 and np.abs(p2 + p0).max() < 1e-6 * max(1.0, np.abs(p0).max())
    osy.set_metric(None)
    with pytest.raises(ValueError):
        make_case("fhn", 6, 4, 2, True, B=1, seed=1, gaussian=True)["osys"].set_metric(M0)


def test_oracle_projection_trace_matches_the_loop_condition():
    """OracleChain.trace (the per-iteration |c|_inf, |delta q|_inf the GPU parity tests consult when an iteration count
    differs): one entry per iteration, the loop condition (sde/mici_extensions.py:1119-1127) holds at the last entry of a
    converged retraction and at no earlier one; a count may only be excused where the disputed entry sits on a tolerance."""
    from test_hip_parity import _count_differs_on_tolerance_edge
    case = make_case("sir", 6, 8, 2, True, B=1, seed=11)
    osy, rng = case["osys"], case["rng"]
    ch = c_oracle.OracleChain(osy)
    ch.set(case["q"][0], rng.standard_normal(osy.Q), case["x_obs"][0], 0)
    ch.project_mom()
    st, itf, itb, _ = ch.step(0.05)
    assert st == 0
    for d, it in ((0, itf), (1, itb)):
        err, ndq = ch.trace(d)
        assert len(err) == it == len(ndq)
        ok = (err < 1e-9) & (ndq < 1e-8)
        assert ok[-1] and not ok[:-1].any()
        # far from the tolerances: no other count can be excused
        assert not _count_differs_on_tolerance_edge(ch, d, it - 1, it)
        # the same trace judged with a tolerance placed on the deciding quantity is an edge
        k = it - 1
        assert _count_differs_on_tolerance_edge(ch, d, k, it, ctol=max(err[k - 1], 1e-300) * 1.001, ptol=1.0)
