"""The Mici-style surface (system.py / integrators.py) against the Python (torch.func) oracle -- which restates the
reference's own classes -- on identical inputs.  Runs on CPU through the TEST-ONLY emulation library (host logic);
tests/test_hip_surface.py repeats the core of it on the GPU."""
import numpy as np
import pytest
from test_emu_logic import emu_lib  # noqa: F401  (fixture)
import manifold_mcmc_for_diffusions_amd as mm
from manifold_mcmc_for_diffusions_amd import example_models as em
from oracle.py import models as omodels, system as osys

TOLS = dict(constraint_tol=1e-9, position_tol=1e-8, max_iters=50)


def build_pair(noisy=True, gaussian=False, T=6, S=4, R=2, seed=3):
    rng = np.random.default_rng(seed)
    sigma = 0.1 if noisy else None
    y = em.simulate_fhn_observations(T, 0.2, 50, seed=seed, sigma=sigma)
    ref = osys.make_system(omodels.fhn, 0.2, S, R, y, sigma=sigma, use_gaussian_splitting=gaussian)
    sysm = mm.ConditionedDiffusionConstrainedSystem(
        0.2, S, R, y, em.fhn.dim_z, em.fhn.dim_x, em.fhn.dim_v, em.fhn.forward_func, em.fhn.generate_x_0,
        em.fhn.generate_z, em.fhn.obs_func, generate_σ=sigma, use_gaussian_splitting=gaussian,
        dim_v_0=em.fhn.dim_v_0)
    gen_init = lambda r: np.concatenate((y, r.standard_normal(y.shape) * 0.5), -1)  # noqa: E731
    ref_state = osys.find_initial_state_by_linear_interpolation(ref, np.random.default_rng(seed + 1), gen_init)
    state = mm.find_initial_state_by_linear_interpolation(sysm, np.random.default_rng(seed + 1), gen_init)
    return ref, sysm, ref_state, state, rng


@pytest.mark.parametrize("noisy,gaussian", [(True, False), (False, False), (True, True)])
def test_initial_state_and_system_methods(emu_lib, noisy, gaussian):  # noqa: F811
    ref, sysm, rs, st, rng = build_pair(noisy, gaussian)
    assert sysm.num_partition == ref.num_partition == 2
    np.testing.assert_allclose(st.pos, rs.pos, rtol=0, atol=1e-11)
    np.testing.assert_allclose(st.mom, rs.mom, rtol=0, atol=1e-9)
    assert np.abs(sysm.constr(st)).max() < 1e-9  # scripts/fhn_model_noisy_obs_chmc_experiment.py:116
    np.testing.assert_allclose(sysm.constr(st), ref.constr(rs), atol=1e-12)
    assert abs(sysm.log_det_sqrt_gram(st) - ref.log_det_sqrt_gram(rs)) < 1e-10
    np.testing.assert_allclose(sysm.grad_log_det_sqrt_gram(st), ref.grad_log_det_sqrt_gram(rs), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sysm.dh1_dpos(st), ref.dh1_dpos(rs), rtol=1e-9, atol=1e-9)
    assert abs(sysm.h(st) - ref.h(rs)) < 1e-8 * max(1.0, abs(ref.h(rs)))
    w = rng.standard_normal(st.pos.shape)
    np.testing.assert_allclose(sysm.normal_space_component(st, w), ref.normal_space_component(rs, w), atol=1e-9)
    chol_C, _ = sysm.chol_gram_blocks(st)
    np.testing.assert_allclose(chol_C, ref.chol_gram_blocks(rs)[0].numpy(), atol=1e-9)


@pytest.mark.parametrize("newton", [True, False])
@pytest.mark.parametrize("gaussian", [False, True])
def test_integrator_step_matches_reference_restatement(emu_lib, newton, gaussian):  # noqa: F811
    ref, sysm, rs, st, rng = build_pair(True, gaussian)
    solver = (mm.jitted_solve_projection_onto_manifold_newton if newton
              else mm.jitted_solve_projection_onto_manifold_quasi_newton)
    rsolver = (osys.jitted_solve_projection_onto_manifold_newton if newton
               else osys.jitted_solve_projection_onto_manifold_quasi_newton)
    integ = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, projection_solver=solver,
                                             projection_solver_kwargs=TOLS)
    rinteg = osys.ConstrainedLeapfrogIntegrator(ref, step_size=0.05, projection_solver=rsolver,
                                                projection_solver_kwargs=TOLS)
    s1, r1 = st, rs
    for _ in range(3):
        s1, r1 = integ.step(s1), rinteg.step(r1)
        np.testing.assert_allclose(s1.pos, r1.pos, rtol=0, atol=1e-9)
        np.testing.assert_allclose(s1.mom, r1.mom, rtol=0, atol=1e-8)
    assert np.abs(sysm.constr(s1)).max() < 1e-9
    # reversibility: flip the direction and step back
    s1.dir = -1
    back = s1
    for _ in range(3):
        back = integ.step(back)
    assert np.abs(back.pos - st.pos).max() < 2e-8


def test_composed_path_equals_fused_path(emu_lib):  # noqa: F811
    _, sysm, _, st, _ = build_pair(True, False)
    fused = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, projection_solver_kwargs=TOLS)
    composed = mm.ConstrainedLeapfrogIntegrator(
        sysm, step_size=0.05, projection_solver_kwargs=TOLS,
        projection_solver=lambda *a, **k: mm.jitted_solve_projection_onto_manifold_newton(*a, **k))
    assert fused._fusable() and not composed._fusable()
    a, b = fused.step(st), composed.step(st)
    np.testing.assert_allclose(a.pos, b.pos, rtol=0, atol=1e-11)
    np.testing.assert_allclose(a.mom, b.mom, rtol=0, atol=1e-10)
    # n_inner_step = 2: fused on the device against the host composition of the same inner steps
    two = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, n_inner_step=2, projection_solver_kwargs=TOLS)
    two_composed = mm.ConstrainedLeapfrogIntegrator(
        sysm, step_size=0.05, n_inner_step=2, projection_solver_kwargs=TOLS,
        projection_solver=lambda *a, **k: mm.jitted_solve_projection_onto_manifold_newton(*a, **k))
    assert two._fusable() and not two_composed._fusable()
    c, d = two.step(st), two_composed.step(st)
    assert np.abs(sysm.constr(c)).max() < 1e-9
    np.testing.assert_allclose(c.pos, d.pos, rtol=0, atol=1e-11)
    np.testing.assert_allclose(c.mom, d.mom, rtol=0, atol=1e-10)


def test_errors_are_the_reference_exceptions(emu_lib):  # noqa: F811
    _, sysm, _, st, _ = build_pair(True, False)
    integ = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=5.0,
                                             projection_solver_kwargs=dict(constraint_tol=1e-9, position_tol=1e-8, max_iters=3))
    with pytest.raises((mm.ConvergenceError, mm.NonReversibleStepError)):
        integ.step(st)
    with pytest.raises(mm.AdaptationError):
        mm.ConstrainedLeapfrogIntegrator(sysm).step(st)
    far = st.copy()
    far.pos = st.pos + 3.0
    with pytest.raises(mm.ConvergenceError):
        mm.jitted_solve_projection_onto_manifold_newton(far, st, 0.1, sysm, max_iters=2)
    with pytest.raises(ValueError):
        mm.ConditionedDiffusionConstrainedSystem(
            0.2, 4, 2, np.zeros((6, 1)), 4, 2, 2, em.fhn.forward_func, em.fhn.generate_x_0, em.fhn.generate_z,
            em.fhn.obs_func, generate_σ=0.1, use_gaussian_splitting=True, metric=object(), dim_v_0=2)
    with pytest.raises(TypeError):
        mm.ConditionedDiffusionConstrainedSystem(0.2, 4, 2, np.zeros((6, 1)), 4, 2, 2, lambda *a: 0, em.fhn.generate_x_0,
                                                 em.fhn.generate_z, em.fhn.obs_func, generate_σ=0.1, dim_v_0=2)


def test_switch_partition_transition(emu_lib):  # noqa: F811
    ref, sysm, rs, st, _ = build_pair(True, False)
    t, rt = mm.SwitchPartitionTransition(sysm), osys.SwitchPartitionTransition(ref)
    s2, _ = t.sample(st.copy())
    r2, _ = rt.sample(rs.copy())
    assert s2.partition == r2.partition == 1
    np.testing.assert_allclose(s2.x_obs_seq, r2.x_obs_seq, atol=1e-12)
    np.testing.assert_allclose(sysm.constr(s2), ref.constr(r2), atol=1e-11)


def test_batched_system(emu_lib):  # noqa: F811
    """B = 3 chains through the same surface: per-chain statuses instead of exceptions."""
    rng = np.random.default_rng(0)
    y = em.simulate_fhn_observations(6, 0.2, 50, seed=5, sigma=0.1)
    sysm = mm.ConditionedDiffusionConstrainedSystem(
        0.2, 4, 2, y, 4, 2, 2, em.fhn.forward_func, em.fhn.generate_x_0, em.fhn.generate_z, em.fhn.obs_func,
        generate_σ=0.1, dim_v_0=2, num_chains=3)
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, 4, y, 3, True, seed=7)
    st = mm.ConditionedDiffusionHamiltonianState(q, xo)
    st.mom = sysm.sample_momentum(st, rng)
    st.dir = np.array([1.0, -1.0, 1.0])
    integ = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, projection_solver_kwargs=TOLS)
    s1 = integ.step(st)
    assert s1.pos.shape == (3, sysm.ctx.Q) and (s1.step_status == 0).all()
    assert np.abs(sysm.constr(s1)).max() < 1e-9


def test_block_metric_surface(emu_lib):  # noqa: F811
    """metric = PositiveDefiniteBlockDiagonalMatrix((DensePositiveDefiniteMatrix(M_0), IdentityMatrix())) as the
    reference accepts it (sde/mici_extensions.py:279-315), set at construction or assigned later (:1926-1931)."""
    from oracle import c_oracle
    _, sysm, _, st, rng = build_pair(True, False)
    a = rng.standard_normal((4, 4))
    M0 = a @ a.T / 4 + 0.5 * np.eye(4)
    metric = mm.PositiveDefiniteBlockDiagonalMatrix((mm.DensePositiveDefiniteMatrix(M0), mm.IdentityMatrix()))
    sysm.metric = metric
    y = sysm.model_dict["y_seq"]
    osy = c_oracle.OracleSystem("fhn", 0.2, 4, 2, y[:, 0], sigma=0.1)
    osy.set_metric(M0)
    st.mom = sysm.sample_momentum(st, np.random.default_rng(2))
    n = np.random.default_rng(2).standard_normal(st.pos.shape)
    n[:4] = np.linalg.cholesky(M0) @ n[:4]  # metric.sqrt @ n (:1257)
    expect = n - osy.jacob_products(st.pos, st.x_obs_seq, st.partition, n, np.zeros(osy.dim_c(st.partition)))[3]
    np.testing.assert_allclose(st.mom, expect, rtol=1e-9, atol=1e-10)
    ch = c_oracle.OracleChain(osy)
    ch.set(st.pos, st.mom, st.x_obs_seq, st.partition)
    assert abs(sysm.h(st) - ch.hamiltonian()) < 1e-9 * max(1.0, abs(ch.hamiltonian()))
    np.testing.assert_allclose(sysm.dh2_dmom(st)[:4], np.linalg.solve(M0, st.mom[:4]), rtol=1e-10)
    np.testing.assert_allclose(sysm.dh2_dmom(st)[4:], st.mom[4:])
    integ = mm.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, projection_solver_kwargs=TOLS)
    new = integ.step(st)
    stt, itf, itb, rev = ch.step(0.05)
    qo, po, _, _ = ch.get()
    assert stt == 0
    np.testing.assert_allclose(new.pos, qo, rtol=0, atol=1e-9 * max(1.0, np.abs(qo).max()))
    np.testing.assert_allclose(new.mom, po, rtol=0, atol=1e-9 * max(1.0, np.abs(po).max()))
    # composed path (python-level A-B-A with the per-op entry points) agrees with the fused one
    composed = mm.ConstrainedLeapfrogIntegrator(
        sysm, step_size=0.05, projection_solver_kwargs=TOLS,
        projection_solver=lambda *a, **k: mm.jitted_solve_projection_onto_manifold_newton(*a, **k))
    b = composed.step(st)
    np.testing.assert_allclose(b.pos, new.pos, rtol=0, atol=1e-10)
    np.testing.assert_allclose(b.mom, new.mom, rtol=0, atol=1e-9)
    # constructor argument, refusals of the reference
    y2 = y
    s2 = mm.ConditionedDiffusionConstrainedSystem(
        0.2, 4, 2, y2, em.fhn.dim_z, em.fhn.dim_x, em.fhn.dim_v, em.fhn.forward_func, em.fhn.generate_x_0,
        em.fhn.generate_z, em.fhn.obs_func, generate_σ=0.1, metric=metric, dim_v_0=em.fhn.dim_v_0)
    assert s2.metric is metric and np.allclose(s2.ctx.M_0, M0)
    with pytest.raises(ValueError):
        mm.ConditionedDiffusionConstrainedSystem(
            0.2, 4, 2, y2, em.fhn.dim_z, em.fhn.dim_x, em.fhn.dim_v, em.fhn.forward_func, em.fhn.generate_x_0,
            em.fhn.generate_z, em.fhn.obs_func, generate_σ=0.1, metric=metric, use_gaussian_splitting=True,
            dim_v_0=em.fhn.dim_v_0)
    with pytest.raises(NotImplementedError):
        sysm.metric = mm.DensePositiveDefiniteMatrix(np.eye(sysm.ctx.Q))
    sysm.metric = None
    assert isinstance(sysm.metric, mm.IdentityMatrix) and sysm.ctx.M_0 is None
