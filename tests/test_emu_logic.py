"""CPU-side tests of the library's HOST logic (kernel sequencing of csrc/chmc_api.inc, masks, status codes,
partition tables, padded layouts) through a TEST-ONLY host emulation build (tests/emu): every device functor is
run in a plain loop.  The emulation library is never loadable through the package (the loader only knows
libchmc_hip.so); it is injected here explicitly.  The GPU parity tests proper are in test_hip_parity.py."""
import ctypes
import os
import subprocess
import numpy as np
import pytest
from helpers import make_case, make_ctx, check_ops_against_oracle, check_steps_against_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu", "libchmc_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    from manifold_mcmc_for_diffusions_amd import _lib
    srcs = [os.path.join(HERE, "emu", f) for f in ("chmc_emu.cpp", "backend_emu.h")]
    csrc = os.path.join(os.path.dirname(HERE), "manifold_mcmc_for_diffusions_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not os.path.exists(EMU) or any(os.path.getmtime(EMU) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I.", "-o", EMU, "chmc_emu.cpp"],
                              cwd=os.path.join(HERE, "emu"))
    saved = _lib._LIB
    _lib._LIB = _lib._bind(ctypes.CDLL(EMU))
    assert _lib.lib().chmc_backend() == b"emu:host-TEST-ONLY"
    yield _lib._LIB
    _lib._LIB = saved


CASES = [
    ("fhn", 6, 4, 2, True, False),
    ("fhn", 7, 5, 3, False, False),
    ("fhn", 6, 4, 2, True, True),
    ("fhn", 5, 4, None, True, False),
    ("fhn", 12, 10, 5, False, True),
    ("sir", 5, 6, None, True, False),
    ("sir", 6, 8, 2, True, False),
    ("sir", 14, 6, 14, True, False),
    ("fhn_nb", 7, 5, 3, False, True),  # the notebook's model: noiseless observations, Gaussian splitting
    ("fhn_nb", 6, 4, 2, True, False),
]


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", CASES)
def test_ops(emu_lib, model, T, S, R, noisy, gaussian):
    case = make_case(model, T, S, R, noisy, B=3, seed=11, gaussian=gaussian)
    ctx = make_ctx(case)
    check_ops_against_oracle(ctx, case)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", CASES)
@pytest.mark.parametrize("newton", [True, False])
def test_steps(emu_lib, model, T, S, R, noisy, gaussian, newton):
    case = make_case(model, T, S, R, noisy, B=4, seed=12, gaussian=gaussian)
    ctx = make_ctx(case)
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05, 0.1, 0.02]), newton=newton, n_steps=3)
    ctx.close()


def test_failed_and_inactive_chains_keep_state(emu_lib):
    case = make_case("fhn", 6, 4, 2, True, B=3, seed=13)
    ctx = make_ctx(case)
    qq = np.repeat(case["q"][:1], 3, 0)
    xx = np.repeat(case["x_obs"][:1], 3, 0)
    ctx.set_state(qq, case["rng"].standard_normal((3, ctx.Q)), xx, 0)
    ctx.project_onto_cotangent_space()
    q0, p0, _, _ = ctx.get_state()
    res = ctx.leapfrog_step(np.array([0.05, 5.0, -0.05]), max_iters=3, active=np.array([1, 1, 0]))
    q1, p1, _, _ = ctx.get_state()
    assert res["status"][0] == 0 and res["status"][1] in (1, 2, 3) and res["status"][2] == -1
    assert np.array_equal(q1[1], q0[1]) and np.array_equal(p1[1], p0[1])
    assert np.array_equal(q1[2], q0[2]) and np.array_equal(p1[2], p0[2])
    assert not np.array_equal(q1[0], q0[0])
    ctx.close()


def test_api_misuse_is_reported(emu_lib):
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    with pytest.raises(RuntimeError, match="sigma"):
        ChmcContext("fhn", 0.2, 4, 2, np.zeros(6), sigma=0.0)
    with pytest.raises(RuntimeError, match="not supported"):
        ChmcContext("fhn", 0.2, 4, None, np.zeros(20), sigma=0.1)  # unpartitioned FHN: 20 rows in one block
    with pytest.raises(RuntimeError, match="at least 2"):  # second partition would open with 1 // 2 = 0 observations
        ChmcContext("fhn", 0.2, 4, 1, np.zeros(6), sigma=0.1)
    from oracle import c_oracle
    with pytest.raises(ValueError):
        c_oracle.OracleSystem("fhn", 0.2, 4, 1, np.zeros(6), sigma=0.1)
    ChmcContext("fhn", 0.2, 4, 1, np.zeros(1), sigma=0.1).close()  # a single observation is a single sub-sequence
    case = make_case("fhn", 6, 4, 2, True, B=2, seed=1)
    ctx = make_ctx(case)
    with pytest.raises(ValueError):
        ctx.set_state(case["q"][:1], None, case["x_obs"], 0)
    with pytest.raises(RuntimeError, match="partition"):
        ctx.set_state(case["q"], None, case["x_obs"], 5)
    with pytest.raises(RuntimeError, match="n_inner_step"):
        ctx.set_state(case["q"], None, case["x_obs"], 0)
        ctx.leapfrog_step(0.1, n_inner_step=0)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 6, 8, 2, True), ("fhn", 5, 7, None, False), ("sir", 6, 5, 3, True)])
def test_init_by_linear_interpolation_matches_host(emu_lib, model, T, S, R, noisy):
    """chmc_init_linear_interpolation against the NumPy restatement of sde/mici_extensions.py:1479-1547 (init.py)."""
    from manifold_mcmc_for_diffusions_amd import example_models as em, init
    case = make_case(model, T, S, R, noisy, B=3, seed=31)
    m = em.MODELS[model]
    rng = np.random.default_rng(7)
    u = case["q"][:, :4].copy()
    v0 = case["q"][:, 4:4 + m.dim_v_0].copy()
    xo = case["x_obs"] + 0.05 * rng.standard_normal(case["x_obs"].shape)
    ctx = make_ctx(case)
    ctx.init_by_linear_interpolation(u, v0, xo, partition=ctx.num_partition - 1)
    q, p, xo_d, part = ctx.get_state()
    assert part == ctx.num_partition - 1 and np.array_equal(xo_d, xo) and not p.any()
    for c in range(3):
        qh, _ = init.find_initial_state_by_linear_interpolation(m, case["obs_interval"], S, case["y"], None,
                                                                lambda r, c=c: xo[c], noisy, u=u[c], v_0=v0[c])
        assert np.abs(q[c] - qh).max() <= 1e-11 * max(1.0, np.abs(qh).max())
    # the interpolated path hits the given states at the observation times
    ctx.update_x_obs_seq()
    assert np.abs(ctx.get_state()[2] - xo).max() <= 1e-9 * (1.0 + np.abs(xo).max())
    ctx.close()



def test_steps_from_unprojected_momentum(emu_lib):
    """The integrator's first half-kick projects whatever momentum it is given (mici _step_a); a momentum set
    without projection must take the full projection path, then the projected-gradient shortcut resumes."""
    case = make_case("fhn", 6, 8, 2, True, B=4, seed=41)
    ctx = make_ctx(case)
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05, 0.1, 0.02]), newton=True, n_steps=3, project=False)
    ctx.close()



def _status_case(emu_lib, ctx_kw, oracle_kw, expect):
    from oracle import c_oracle
    case = make_case("fhn", 6, 8, 2, True, B=3, seed=51)
    ctx = make_ctx(case)
    B = 3
    qq = np.repeat(case["q"][:1], B, 0)
    xx = np.repeat(case["x_obs"][:1], B, 0)
    ctx.set_state(qq, case["rng"].standard_normal((B, ctx.Q)), xx, 0)
    ctx.project_onto_cotangent_space()
    q0, p0, _, _ = ctx.get_state()
    dts = np.array([0.05, -0.08, 0.1])
    res = ctx.leapfrog_step(dts, **ctx_kw)
    q1, p1, _, _ = ctx.get_state()
    for c in range(B):
        ch = c_oracle.OracleChain(case["osys"])
        ch.set(qq[c], p0[c], xx[c], 0)
        st, itf, itb, rev = ch.step(dts[c], **oracle_kw)
        assert res["status"][c] == st == expect, (c, res["status"][c], st)
        assert np.array_equal(q1[c], q0[c]) and np.array_equal(p1[c], p0[c])  # a failed step leaves the state alone
        if expect == 3:
            assert res["iters_fwd"][c] == itf and 0 < res["rev_err"][c] < 1e-10 and 0 < rev < 1e-10  # both round-off
    ctx.close()


def test_status_diverged_matches_oracle(emu_lib):
    """divergence_tol below the first constraint error: 'iteration diverged' (sde/mici_extensions.py:1393-1397)."""
    _status_case(emu_lib, dict(divergence_tol=1e-12), dict(dtol=1e-12), 2)


def test_status_not_converged_matches_oracle(emu_lib):
    """max_iters too small: 'did not converge' (:1398-1402)."""
    _status_case(emu_lib, dict(max_iters=1), dict(max_iters=1), 1)


def test_status_non_reversible_matches_oracle(emu_lib):
    """reverse_check_tol below the round-off of the forward-backward retraction: NonReversibleStepError (mici)."""
    _status_case(emu_lib, dict(reverse_check_tol=1e-22), dict(rev_tol=1e-22), 3)


METRIC_CASES = [("fhn", 6, 4, 2, True), ("fhn", 7, 5, 3, False), ("sir", 6, 8, 2, True), ("sir", 14, 6, 14, True),
                ("fhn_nb", 6, 4, 2, True)]


@pytest.mark.parametrize("model,T,S,R,noisy", METRIC_CASES)
@pytest.mark.parametrize("newton", [True, False])
def test_block_metric(emu_lib, model, T, S, R, noisy, newton):
    """M = blockdiag(M_0, I) on the u-part (sde/mici_extensions.py:279-315, 794-798, 1033-1041, 1105-1113, 1202-1259)."""
    from helpers import check_block_metric_against_oracle
    case = make_case(model, T, S, R, noisy, B=4, seed=41)
    ctx = make_ctx(case)
    check_block_metric_against_oracle(ctx, case, newton, np.array([0.05, -0.05, 0.08, 0.02]))
    ctx.close()


def test_block_metric_misuse(emu_lib):
    case = make_case("fhn", 6, 4, 2, True, B=2, seed=42, gaussian=True)
    ctx = make_ctx(case)
    with pytest.raises(RuntimeError, match="Gaussian splitting"):
        ctx.set_metric(np.eye(4) * 2.0)
    ctx.close()
    case = make_case("fhn", 6, 4, 2, True, B=2, seed=42)
    ctx = make_ctx(case)
    with pytest.raises(RuntimeError, match="positive definite"):
        ctx.set_metric(-np.eye(4))
    bad = np.eye(4)
    bad[0, 1] = 0.3
    with pytest.raises(RuntimeError, match="symmetric"):
        ctx.set_metric(bad)
    with pytest.raises(ValueError):
        ctx.set_metric(np.eye(3))
    ctx.close()


@pytest.mark.parametrize("with_metric", [False, True])
@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 6, 4, 2, True), ("fhn", 7, 5, 3, False)])  # even and odd dim_q
def test_tree_leaf(emu_lib, model, T, S, R, noisy, with_metric):
    from helpers import check_tree_leaf
    case = make_case(model, T, S, R, noisy, B=5, seed=51)
    ctx = make_ctx(case)
    check_tree_leaf(ctx, case, "cpu", with_metric)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,newton", [
    ("fhn", 12, 10, 5, True, False, True), ("fhn", 7, 5, 3, False, True, False), ("sir", 6, 8, 2, True, False, True)])
def test_half_batches_equal_one_batch(emu_lib, monkeypatch, model, T, S, R, noisy, gaussian, newton):
    """Host logic of the two-half-batch step (chain-range views, lock-step Newton loops, per-half counters) against the
    one-batch step: bitwise equal, masked and failing chains in both halves."""
    from helpers import halves_vs_single_batch
    case = make_case(model, T, S, R, noisy, B=7, seed=81, gaussian=gaussian)
    for part in range(2):
        halves_vs_single_batch(case, monkeypatch, part=part, newton=newton, masked=(1, 6), failing=(2, 4))


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,newton,n_inner", [
    ("fhn", 6, 4, 2, True, False, True, 2), ("fhn", 7, 5, 3, False, True, True, 3), ("fhn", 12, 10, 5, True, False, False, 2),
    ("sir", 6, 8, 2, True, False, True, 2)])
def test_inner_h2_flow_steps(emu_lib, model, T, S, R, noisy, gaussian, newton, n_inner):
    """n_inner_step > 1 on the fused path (mici _step_b loop; scripts/utils.py:131-136 --num-inner-h2-step) against the
    C oracle, including a chain that fails in a LATER inner step and must get its start state back."""
    case = make_case(model, T, S, R, noisy, B=4, seed=91, gaussian=gaussian)
    ctx = make_ctx(case)
    check_steps_against_oracle(ctx, case, np.array([0.06, -0.06, 0.1, 0.03]), newton=newton, n_steps=2, n_inner=n_inner)
    check_steps_against_oracle(ctx, case, np.array([0.06, -0.06, 0.1, 0.03]), newton=newton, n_steps=1, n_inner=n_inner,
                               project=False)
    ctx.close()


def test_failure_in_a_later_inner_step_restores_the_start_state(emu_lib):
    from helpers import check_late_inner_failure
    found = 0
    for seed in range(93, 101):  # (which chain has the larger round-off in its last inner step depends on the arithmetic)
        case = make_case("fhn", 12, 10, 5, True, B=6, seed=seed)
        ctx = make_ctx(case)
        found += check_late_inner_failure(ctx, case, np.array([0.08, -0.08, 0.1, 0.05, -0.1, 0.07])) is not None
        ctx.close()
        if found >= 2:
            break
    assert found >= 1


# variable observation noise: sigma = generate_σ_y(u) = exp(u[dim_z]), dim_u = dim_z + 1
# (sde/mici_extensions.py:353-358, 559-569, 601-608; scripts/sir_model_chmc_experiment.py:44,58,77)
VS_CASES = [("sir", 5, 6, None, False), ("sir", 6, 8, 2, False), ("fhn", 6, 4, 2, False), ("fhn", 7, 5, 3, True),
            ("sir", 14, 6, 14, False)]


@pytest.mark.parametrize("model,T,S,R,gaussian", VS_CASES)
def test_variable_observation_noise(emu_lib, model, T, S, R, gaussian):
    case = make_case(model, T, S, R, True, B=3, seed=11, gaussian=gaussian, var_sigma=True)
    ctx = make_ctx(case)
    assert ctx.U == 5 and ctx.Q == case["q"].shape[1]
    check_ops_against_oracle(ctx, case)
    ctx.close()
    case = make_case(model, T, S, R, True, B=4, seed=12, gaussian=gaussian, var_sigma=True)
    ctx = make_ctx(case)
    dts = np.array([0.05, -0.05, 0.1, 0.02])
    for newton in ((True,) if R == 14 else (True, False)):  # (14-row quasi-Newton counts sit on the tolerance's edge)
        for part in range(ctx.num_partition):
            check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=2, part=part)
    check_steps_against_oracle(ctx, case, dts, n_steps=1, project=False)
    check_steps_against_oracle(ctx, case, dts, n_steps=1, n_inner=2)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R", [("sir", 6, 8, 2), ("fhn", 6, 4, 2)])
def test_variable_observation_noise_block_metric(emu_lib, model, T, S, R):
    """Odd dim_u = 5 with M = blockdiag(M_0, I)."""
    from helpers import check_block_metric_against_oracle
    case = make_case(model, T, S, R, True, B=4, seed=41, var_sigma=True)
    ctx = make_ctx(case)
    check_block_metric_against_oracle(ctx, case, True, np.array([0.05, -0.05, 0.08, 0.02]))
    ctx.close()


def test_adam_update_entry_point(emu_lib):
    """chmc_adam_update_device (the Adam step of a device-resident iteration of
    find_initial_state_by_gradient_descent_noisy_system, sde/mici_extensions.py:1679-1801) on the host emulation, where
    "device" buffers are host arrays: equal to NumPy, a non-finite gradient entry counts as zero in the moments and a
    zero learning rate keeps a chain's parameters.  (The objective half, chmc_adam_objective_device, runs wave kernels: GPU
    test in test_hip_surface.py.)"""
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    counts = np.array([3.0, 8.0, 28.0, 75.0, 221.0])
    B, T, S = 5, len(counts), 6
    ctx = ChmcContext("sir", 1.0, S, T, counts, sigma=1.0, num_chains=B)
    n = ctx.Q - T
    rng = np.random.default_rng(8)
    u_v = np.ascontiguousarray(0.4 * rng.standard_normal((B, n)))
    g = rng.standard_normal((B, n))
    g[1, 3] = np.nan
    m0, v0 = rng.standard_normal((B, n)), rng.random((B, n))
    m, v, u = m0.copy(), v0.copy(), u_v.copy()
    tt = np.array([1.0, 2.0, 5.0, 9.0, 30.0])
    lr = np.array([0.1, 0.1, 0.0, 0.1, 0.1]) / (1 - 0.9 ** tt)
    coef = np.stack([1.0 / (1 - 0.999 ** tt), lr], 1)
    ctx.adam_update_device(u.ctypes.data, m.ctypes.data, v.ctypes.data, g.ctypes.data, coef, 0.9, 0.999, 1e-8)
    g0 = np.where(np.isfinite(g), g, 0.0)
    m1, v1 = 0.9 * m0 + (1 - 0.9) * g0, 0.999 * v0 + (1 - 0.999) * g0 ** 2
    np.testing.assert_allclose(m, m1, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(v, v1, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(u, u_v - lr[:, None] * m1 / (np.sqrt(v1 * coef[:, :1]) + 1e-8), rtol=1e-13, atol=1e-15)
    assert np.array_equal(u[2], u_v[2])
    with pytest.raises(RuntimeError):
        ctx.adam_objective_device(u.ctypes.data, g.ctypes.data)   # wave kernels only: reported, not emulated
    ctx.close()
