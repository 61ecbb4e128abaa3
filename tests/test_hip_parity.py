"""GPU parity tests: the HIP library (through its C ABI) against the C oracle on identical seeded inputs.

Tolerances: the path is fp64 end to end and both sides use the same generated model arithmetic, so the only
differences are summation order (Gram / dot-product reductions) and FMA contraction; per-op results must agree
to 1e-10 relative (inf-norm), single leapfrog steps to 1e-9 relative at identical Newton iteration counts
(SURVEY.md section 8c)."""
import os
import sys
import numpy as np
import pytest
from helpers import make_case, make_ctx, check_ops_against_oracle, check_steps_against_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SMALL = [
    # model, T, S, R, noisy, gaussian
    ("fhn", 6, 4, 2, True, False),
    ("fhn", 7, 5, 3, False, False),
    ("fhn", 6, 4, 2, True, True),
    ("fhn", 5, 4, None, True, False),
    ("fhn", 12, 10, 5, True, False),
    ("fhn", 12, 10, 5, False, True),
    ("sir", 5, 6, None, True, False),
    ("sir", 6, 8, 2, True, False),
    ("sir", 14, 6, 14, True, False),
    ("fhn_nb", 7, 5, 3, False, True),  # the notebook's model: noiseless observations, Gaussian splitting
    ("fhn_nb", 6, 4, 2, True, False),
]


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", SMALL)
def test_ops_small(model, T, S, R, noisy, gaussian):
    case = make_case(model, T, S, R, noisy, B=3, seed=11, gaussian=gaussian)
    ctx = make_ctx(case)
    assert ctx.L.chmc_backend() == b"hip:gfx950"
    check_ops_against_oracle(ctx, case)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", SMALL)
@pytest.mark.parametrize("newton", [True, False])
def test_steps_small(model, T, S, R, noisy, gaussian, newton):
    case = make_case(model, T, S, R, noisy, B=4, seed=12, gaussian=gaussian)
    ctx = make_ctx(case)
    dts = np.array([0.05, -0.05, 0.1, 0.02])
    check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=3)
    ctx.close()


def test_failed_chains_keep_state():
    case = make_case("fhn", 6, 4, 2, True, B=3, seed=13)
    ctx = make_ctx(case)
    dts = np.array([0.05, 5.0, -0.05])  # the middle chain cannot converge with max_iters=3
    qq = np.repeat(case["q"][:1], 3, 0)
    xx = np.repeat(case["x_obs"][:1], 3, 0)
    ctx.set_state(qq, case["rng"].standard_normal((3, ctx.Q)), xx, 0)
    ctx.project_onto_cotangent_space()
    q0, p0, _, _ = ctx.get_state()
    res = ctx.leapfrog_step(dts, max_iters=3, active=np.array([1, 1, 0]))
    q1, p1, _, _ = ctx.get_state()
    assert res["status"][0] == 0 and res["status"][1] in (1, 2, 3) and res["status"][2] == -1
    assert np.array_equal(q1[1], q0[1]) and np.array_equal(p1[1], p0[1])
    assert np.array_equal(q1[2], q0[2]) and np.array_equal(p1[2], p0[2])
    assert not np.array_equal(q1[0], q0[0])
    ctx.close()


def test_baseline_config1_fhn_noisy_s50():
    """BASELINE.json configs[0]: FHN noisy-obs, 50 inter-obs steps (T=100, R=5), here 2 chains."""
    case = make_case("fhn", 100, 50, 5, True, B=2, seed=14)
    ctx = make_ctx(case)
    check_ops_against_oracle(ctx, case)
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05]), n_steps=2)
    ctx.close()


def test_baseline_config2_size_fhn_noisy_s400():
    """Full-size shape of BASELINE.json configs[1] (Q = 80106), 2 chains, one step against the oracle."""
    case = make_case("fhn", 100, 400, 5, True, B=2, seed=15)
    ctx = make_ctx(case)
    assert ctx.Q == 80106 and ctx.C == [138, 140] and ctx.K == [20, 21]
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05]), n_steps=1)
    ctx.close()


def test_baseline_config3_size_fhn_noiseless_s400():
    case = make_case("fhn", 100, 400, 5, False, B=2, seed=16)
    ctx = make_ctx(case)
    assert ctx.Q == 80006 and ctx.C == [119, 120]
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05]), n_steps=1)
    ctx.close()


def test_baseline_config4_size_sir_s200():
    """Full-size shape of BASELINE.json configs[3]: SIR, T = 14, S = 200, R = 14 (one dense 14-row block, Q = 8419);
    synthetic data from the prior (the boarding-school counts need the Adam initialiser)."""
    case = make_case("sir", 14, 200, 14, True, B=2, seed=19, obs_interval=0.25)
    ctx = make_ctx(case)
    assert ctx.Q == 8419 and ctx.C == [14] and ctx.K == [1] and ctx.RM == 16
    check_ops_against_oracle(ctx, case, tol=1e-9)
    check_steps_against_oracle(ctx, case, np.array([0.02, -0.02]), n_steps=1)
    ctx.close()


def test_baseline_config5_size_fhn_noisy_s800():
    """Full-size shape of BASELINE.json configs[4] (S = 800, Q = 160106), 2 chains, one step against the oracle."""
    case = make_case("fhn", 100, 800, 5, True, B=2, seed=20)
    ctx = make_ctx(case)
    assert ctx.Q == 160106 and ctx.C == [138, 140]
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05]), n_steps=1)
    ctx.close()


def test_size_independent_properties_full_size():
    """At full size: after a successful step |c|_inf < ctol, J p = 0 (momentum tangent), and a step followed by a
    direction flip returns to the start (reversibility, tolerance 2e-8 as reverse_check_tol)."""
    case = make_case("fhn", 100, 400, 5, True, B=4, seed=17)
    ctx = make_ctx(case)
    B = 4
    qq = np.repeat(case["q"][:1], B, 0)
    xx = np.repeat(case["x_obs"][:1], B, 0)
    ctx.set_state(qq, case["rng"].standard_normal((B, ctx.Q)), xx, 0)
    ctx.project_onto_cotangent_space()
    q0, p0, _, _ = ctx.get_state()
    dt = np.array([0.05, 0.03, -0.04, 0.02])
    res = ctx.leapfrog_step(dt)
    assert (res["status"] == 0).all()
    assert np.abs(ctx.constr()).max() < 1e-9
    q1, p1, _, _ = ctx.get_state()
    Jp = ctx.lmult_by_jacob_constr(p1)
    assert np.abs(Jp).max() < 1e-9 * np.abs(p1).max() * np.sqrt(ctx.Q)
    res2 = ctx.leapfrog_step(-dt)
    assert (res2["status"] == 0).all()
    q2, p2, _, _ = ctx.get_state()
    assert np.abs(q2 - q0).max() < 2e-8
    assert np.abs(p2 - p0).max() < 1e-6
    ctx.close()


def test_switch_partition_matches_oracle():
    from oracle import c_oracle
    case = make_case("fhn", 12, 10, 5, True, B=2, seed=18)
    ctx = make_ctx(case)
    ctx.set_state(case["q"], None, case["x_obs"], 0)
    ctx.switch_partition()
    _, _, xo, part = ctx.get_state()
    assert part == 1
    for c in range(2):
        ch = c_oracle.OracleChain(case["osys"])
        ch.set(case["q"][c], None, case["x_obs"][c], 0)
        ch.switch_partition()
        _, _, xo_o, part_o = ch.get()
        assert part_o == 1
        assert np.abs(xo[c] - xo_o).max() < 1e-12
        assert abs(ctx.log_det_sqrt_gram()[c] - ch.log_det()) < 1e-10
    ctx.close()


# ---- the hand-scheduled forward scan (k_fwd_scan) runs when the steps per observation are a multiple of 8
SCAN = [
    # model, T, S, R, noisy, gaussian
    ("fhn", 6, 8, 2, True, False),
    ("fhn", 7, 16, 3, False, False),
    ("fhn", 6, 8, 2, True, True),
    ("sir", 14, 8, 14, True, False),
    ("sir", 6, 16, 2, True, False),
    ("fhn_nb", 7, 8, 3, False, True),
]


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", SCAN)
def test_ops_forward_scan_kernel(model, T, S, R, noisy, gaussian):
    """23 chains: B * K is neither a multiple of 64 nor below it for the partitioned cases, so full, partial and
    (for K = 1) single workgroups of the scan are all exercised."""
    case = make_case(model, T, S, R, noisy, B=23, seed=21, gaussian=gaussian)
    ctx = make_ctx(case)
    check_ops_against_oracle(ctx, case)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", SCAN)
def test_steps_forward_scan_kernel(model, T, S, R, noisy, gaussian):
    case = make_case(model, T, S, R, noisy, B=5, seed=22, gaussian=gaussian)
    ctx = make_ctx(case)
    dts = np.array([0.05, -0.05, 0.1, 0.02, -0.08])
    check_steps_against_oracle(ctx, case, dts, newton=True, n_steps=3)
    ctx.close()


def test_forward_scan_kernel_matches_generic_functor(monkeypatch):
    """Same inputs through both forward-scan implementations: constraint values, Jacobian (which is built from the
    stored trajectory) and two leapfrog steps with a masked chain.  The arithmetic per step is the same generated
    code, so only FMA contraction can differ: 1e-13 relative."""
    case = make_case("fhn", 9, 16, 3, True, B=70, seed=23)
    rng = np.random.default_rng(5)
    p = rng.standard_normal(case["q"].shape)
    qq = np.repeat(case["q"][:1], 70, 0)
    xx = np.repeat(case["x_obs"][:1], 70, 0)
    dts = 0.02 + 0.08 * rng.random(70)
    act = np.ones(70, dtype=np.int32)
    act[[3, 64, 69]] = 0
    out = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("CHMC_NO_FWD_SCAN", "1")
        else:
            monkeypatch.delenv("CHMC_NO_FWD_SCAN", raising=False)
        ctx = make_ctx(case)
        ctx.set_state(case["q"], p, case["x_obs"], 1)
        c = ctx.constr()
        du, dv = ctx.jacob_constr_blocks()
        ctx.set_state(qq, p, xx, 0)
        ctx.project_onto_cotangent_space()
        res = [ctx.leapfrog_step(dts, active=act) for _ in range(2)]
        q1, p1, _, _ = ctx.get_state()
        out.append((c, du, dv, q1, p1, res))
        ctx.close()
    a, b = out
    for x, y in zip(a[:5], b[:5]):
        assert np.abs(x - y).max() <= 1e-13 * max(1.0, np.abs(y).max())
    for ra, rb in zip(a[5], b[5]):
        assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["iters_fwd"], rb["iters_fwd"])
        assert (ra["status"][[3, 64, 69]] == -1).all()


@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 6, 8, 2, True), ("fhn", 5, 7, None, False), ("sir", 6, 5, 3, True)])
def test_device_initial_states_match_host(model, T, S, R, noisy):
    """chmc_init_linear_interpolation (sde/mici_extensions.py:1479-1547, batched) against init.py's NumPy version,
    and the manifold property of the result: the path interpolates the given states."""
    from manifold_mcmc_for_diffusions_amd import example_models as em, init
    case = make_case(model, T, S, R, noisy, B=5, seed=31)
    m = em.MODELS[model]
    rng = np.random.default_rng(7)
    u = case["q"][:, :4].copy()
    v0 = case["q"][:, 4:4 + m.dim_v_0].copy()
    xo = case["x_obs"] + 0.05 * rng.standard_normal(case["x_obs"].shape)
    ctx = make_ctx(case)
    ctx.init_by_linear_interpolation(u, v0, xo, partition=0)
    q, p, xo_d, part = ctx.get_state()
    assert part == 0 and np.array_equal(xo_d, xo) and not p.any()
    for c in range(5):
        qh, _ = init.find_initial_state_by_linear_interpolation(m, case["obs_interval"], S, case["y"], None,
                                                                lambda r, c=c: xo[c], noisy, u=u[c], v_0=v0[c])
        assert np.abs(q[c] - qh).max() <= 1e-11 * max(1.0, np.abs(qh).max())
    # state rows of the constraint (x at the sub-sequence ends minus x_obs_seq) vanish on the interpolated path
    ctx.update_x_obs_seq()
    _, _, xo_full, _ = ctx.get_state()
    assert np.abs(xo_full - xo).max() <= 1e-9 * (1.0 + np.abs(xo).max())
    ctx.close()


def test_workload_device_init_equals_host_init():
    from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
    a = FhnWorkload(num_chains=3, num_steps_per_obs=16, num_obs=10, num_steps_per_obs_data=200)
    b = FhnWorkload(num_chains=3, num_steps_per_obs=16, num_obs=10, num_steps_per_obs_data=200, device_init=True)
    qa, _, xa, _ = a.ctx.get_state()
    qb, _, xb, _ = b.ctx.get_state()
    assert np.array_equal(xa, xb) and np.abs(qa - qb).max() <= 1e-11 * max(1.0, np.abs(qa).max())
    assert np.abs(a.ctx.hamiltonian() - b.ctx.hamiltonian()).max() <= 1e-7 * np.abs(a.ctx.hamiltonian()).max()
    a.ctx.close(), b.ctx.close()



def test_steps_from_unprojected_momentum():
    """The integrator's first half-kick projects whatever momentum it is given (mici _step_a); a momentum set
    without projection must take the full projection path, then the projected-gradient shortcut resumes."""
    case = make_case("fhn", 6, 8, 2, True, B=4, seed=41)
    ctx = make_ctx(case)
    check_steps_against_oracle(ctx, case, np.array([0.05, -0.05, 0.1, 0.02]), newton=True, n_steps=3, project=False)
    ctx.close()



def _status_case(ctx_kw, oracle_kw, expect):
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


def test_status_diverged_matches_oracle():
    """divergence_tol below the first constraint error: 'iteration diverged' (sde/mici_extensions.py:1393-1397)."""
    _status_case(dict(divergence_tol=1e-12), dict(dtol=1e-12), 2)


def test_status_not_converged_matches_oracle():
    """max_iters too small: 'did not converge' (:1398-1402)."""
    _status_case(dict(max_iters=1), dict(max_iters=1), 1)


def test_status_non_reversible_matches_oracle():
    """reverse_check_tol below the round-off of the forward-backward retraction: NonReversibleStepError (mici)."""
    _status_case(dict(reverse_check_tol=1e-22), dict(rev_tol=1e-22), 3)


def test_sir_initial_states_by_gradient_descent():
    """The SIR script's initialisation (find_initial_state_by_gradient_descent_noisy_system, :1679-1801), batched on
    top of the library's constraint / adjoint operators: found states lie on the manifold, the residuals became the
    observation-noise components, and the gradient used by Adam agrees with finite differences."""
    from manifold_mcmc_for_diffusions_amd import example_models as em, init
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    T, S, B = 6, 8, 5
    rng = np.random.default_rng(3)
    y = np.array([3.0, 8.0, 28.0, 75.0, 120.0, 170.0])  # infection counts rising as in the boarding-school data
    ctx = ChmcContext("sir", 1.0, S, T, y, sigma=1.0, num_chains=B)
    assert ctx.num_blocks == 1
    # gradient check of the objective at a random point
    u_v = 0.3 * rng.standard_normal((B, ctx.Q - T))
    u_v[:, :4] += np.array([-1.0, -1.0, 1.0, 0.0])

    def objective(uv):
        ctx.set_state(np.concatenate([uv, np.zeros((B, T))], 1), None, np.zeros((B, T, 3)), 0)
        c = ctx.constr()
        return 0.5 * (c ** 2).sum(1) + 0.5 * (uv ** 2).sum(1), c

    f0, c0 = objective(u_v)
    g = ctx.rmult_by_jacob_constr(c0)[:, :ctx.Q - T] + u_v
    for k in (0, 2, 4, 7, 40):
        e = np.zeros_like(u_v)
        e[:, k] = 1e-6
        fd = (objective(u_v + e)[0] - objective(u_v - e)[0]) / 2e-6
        assert np.abs(fd - g[:, k]).max() <= 1e-5 * max(1.0, np.abs(g[:, k]).max())
    q, xo, tries = init.find_initial_states_by_gradient_descent_noisy_system(ctx, rng, adam_step_size=1e-1,
                                                                             max_iters=5000)
    assert np.abs(ctx.constr()).max() < 1e-9            # on the manifold
    assert (np.mean(q[:, -T:] ** 2, 1) < 1.0).all()      # mean squared residual below the threshold
    assert np.isfinite(ctx.hamiltonian()).all() and (tries >= 1).all()
    ctx.sample_momentum(1, 1)
    r = ctx.leapfrog_step(np.full(B, 0.02))
    assert (r["status"] == 0).mean() >= 0.6
    ctx.close()


@pytest.mark.parametrize("model,T,S,gaussian", [("fhn", 5, 7, False), ("fhn", 3, 70, True), ("sir", 4, 9, False),
                                                ("fhn", 4, 16, False), ("sir", 3, 8, True)])  # S % 8 == 0: scan kernel
def test_unconstrained_hmc_target_matches_autodiff_oracle(model, T, S, gaussian):
    """chmc_neg_log_dens_and_grad (conditioned_diffusion_neg_log_dens_and_grad, sde/mici_extensions.py:82-205: the
    reference's unconstrained-HMC comparator) against the torch autograd restatement: value and gradient."""
    from oracle.py.neg_log_dens import neg_log_dens_and_grad
    case = make_case(model, T, S, None, True, B=3, seed=61)
    ctx = make_ctx(case)
    QH = ctx.U + ctx.NV
    q = case["q"][:, :QH]
    val, g = ctx.neg_log_dens_and_grad(q, use_gaussian_splitting=gaussian)
    for c in range(3):
        vo, go = neg_log_dens_and_grad(model, case["obs_interval"], S, case["y"], case["sigma"], q[c], gaussian)
        assert abs(val[c] - vo) <= 1e-10 * max(1.0, abs(vo))
        assert np.abs(g[c] - go).max() <= 1e-9 * max(1.0, np.abs(go).max())
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 6, 4, 2, True), ("fhn", 12, 16, 5, True), ("fhn", 7, 8, 3, False),
                                               ("sir", 6, 8, 2, True), ("sir", 14, 6, 14, True), ("fhn_nb", 6, 4, 2, True)])
@pytest.mark.parametrize("newton", [True, False])
def test_block_metric(model, T, S, R, noisy, newton):
    """M = blockdiag(M_0, I) on the u-part (sde/mici_extensions.py:279-315, 794-798, 1033-1041, 1105-1113, 1202-1259):
    per-op entry points, retraction with its multiplier term, momentum sampling and fused steps against the oracle."""
    from helpers import check_block_metric_against_oracle
    case = make_case(model, T, S, R, noisy, B=4, seed=41)
    ctx = make_ctx(case)
    check_block_metric_against_oracle(ctx, case, newton, np.array([0.05, -0.05, 0.08, 0.02]))
    ctx.close()


@pytest.mark.parametrize("with_metric", [False, True])
@pytest.mark.parametrize("model,T,S,R,noisy", [("fhn", 12, 16, 5, True), ("fhn", 7, 5, 3, False), ("sir", 6, 8, 2, True)])
def test_tree_leaf(model, T, S, R, noisy, with_metric):
    """chmc_tree_leaf (per-leaf bookkeeping of the batched no-U-turn trees) against numpy on the same device buffers."""
    from helpers import check_tree_leaf
    case = make_case(model, T, S, R, noisy, B=5, seed=51)
    ctx = make_ctx(case)
    check_tree_leaf(ctx, case, "cuda", with_metric)
    ctx.close()


def _distinct_on_manifold_chains(model, T, S, R, B, seed, obs_interval=None):
    """B DIFFERENT chain states that all lie on the constraint manifold of one noisy-observation data set: every chain
    has its own (u, v_0, v_seq); x_obs_seq is its own trajectory (state rows vanish) and its observation-noise
    components are solved for, n_t = (y_t - obs_func(x_t)) / sigma (observation rows vanish)."""
    from manifold_mcmc_for_diffusions_amd import example_models as em
    case = make_case(model, T, S, R, True, B=B, seed=seed, obs_interval=obs_interval)
    m, q, xo, y, sigma = em.MODELS[model], case["q"], case["x_obs"], case["y"], case["sigma"]
    q[:, -T:] = (y[None, :] - m.obs_func(xo)[..., 0]) / sigma
    return case


def test_full_size_distinct_chains_with_masked_subset():
    """BASELINE.json configs[1] at full size (Q = 80106) with 72 DISTINCT chains (more than one wavefront of chains,
    B K = 1440 / 1512 blocks: full and partial wavefronts of every kernel), a subset masked out by `active`, two chains
    whose retraction cannot converge (masked inside the Newton loop from then on) -- every chain against the C oracle,
    in both partitions."""
    from oracle import c_oracle
    B = 72
    case = _distinct_on_manifold_chains("fhn", 100, 400, 5, B, seed=71)
    ctx = make_ctx(case)
    assert ctx.Q == 80106
    rng = case["rng"]
    inactive, failing = [5, 17, 33, 64, 71], [7, 40]
    for part in (0, 1):
        ctx.set_state(case["q"], rng.standard_normal((B, ctx.Q)), case["x_obs"], part)
        assert np.abs(ctx.constr()).max() < 1e-9
        ctx.project_onto_cotangent_space()
        q0, p0, _, _ = ctx.get_state()
        dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.02 + 0.04 * rng.random(B))
        dts[failing] = 5.0
        act = np.ones(B, dtype=np.int32)
        act[inactive] = 0
        res = ctx.leapfrog_step(dts, active=act, max_iters=12)
        q1, p1, _, _ = ctx.get_state()
        assert (res["status"][inactive] == -1).all() and (res["status"][failing] > 0).all()
        n_ok = 0
        for c in range(B):
            if not act[c]:
                assert np.array_equal(q1[c], q0[c]) and np.array_equal(p1[c], p0[c])
                continue
            ch = c_oracle.OracleChain(case["osys"])
            ch.set(case["q"][c], p0[c], case["x_obs"][c], part)
            st, itf, itb, _ = ch.step(dts[c], max_iters=12)
            qo, po, _, _ = ch.get()
            assert res["status"][c] == st, (part, c, res["status"][c], st)
            assert res["iters_fwd"][c] == itf and (st != 0 or res["iters_bwd"][c] == itb), (part, c)
            assert np.abs(q1[c] - qo).max() <= 1e-9 * max(1.0, np.abs(qo).max()), (part, c)
            assert np.abs(p1[c] - po).max() <= 1e-9 * max(1.0, np.abs(po).max()), (part, c)
            n_ok += st == 0
        assert n_ok >= B - len(inactive) - len(failing) - 3
    ctx.close()


def _count_differs_on_tolerance_edge(ch, direction, it_lib, it_orc, ctol=1e-9, ptol=1e-8, rel=1e-2):
    """A retraction's iteration count may differ from the oracle's only where the oracle's own deciding quantity of the
    loop condition (sde/mici_extensions.py:1119-1127: |c|_inf < ctol and |delta q|_inf < ptol) sits within `rel`
    (relative) of its tolerance AT THE DISPUTED COUNT, i.e. after min(it_lib, it_orc) iterations: the side that stopped
    there saw the test pass, the other saw it fail, and an evaluation order that moves the quantity by that little
    decides either way."""
    err, ndq = ch.trace(direction)
    k = min(it_lib, it_orc)
    if k < 1 or k > len(err):
        return False
    e, d = err[k - 1], ndq[k - 1]
    if not (np.isfinite(e) and np.isfinite(d)):
        return False
    near_c, near_p = abs(e - ctol) <= rel * ctol, abs(d - ptol) <= rel * ptol
    # the condition that is not on the edge must hold outright, otherwise nothing could have stopped at k
    return (near_c and (d < ptol or near_p)) or (near_p and (e < ctol or near_c))


def _step_every_chain_against_oracle(ctx, osys, q, p0, xo, part, dts, act, max_iters, tol=1e-9, edge_ok=False):
    """One batched leapfrog step from the given per-chain states against one oracle chain per chain.
    edge_ok: layouts whose forward scan is the time-parallel one (its constraint values agree with the sequential
    recursion to about 1e-12, not bitwise) may differ in a Newton iteration count where the oracle's residual lies within
    1e-2 (relative) of a convergence tolerance at that count; every other layout gets no allowance."""
    from oracle import c_oracle
    B = len(q)
    res = ctx.leapfrog_step(dts, active=act, max_iters=max_iters)
    q1, p1, _, _ = ctx.get_state()
    n_ok, edge = 0, []
    for c in range(B):
        if not act[c]:
            assert res["status"][c] == -1 and np.array_equal(q1[c], q[c]) and np.array_equal(p1[c], p0[c]), (part, c)
            continue
        ch = c_oracle.OracleChain(osys)
        ch.set(q[c], p0[c], xo[c], part)
        st, itf, itb, _ = ch.step(dts[c], max_iters=max_iters)
        qo, po, _, _ = ch.get()
        assert res["status"][c] == st, (part, c, res["status"][c], st)
        same_f = res["iters_fwd"][c] == itf
        same_b = st != 0 or res["iters_bwd"][c] == itb
        ctol = tol
        if not (same_f and same_b):
            got = (int(res["iters_fwd"][c]), int(res["iters_bwd"][c]))
            assert edge_ok and st == 0, (part, c, got, (itf, itb))
            # (a different forward count moves the new point by < position_tol, so the reverse counts are only comparable
            # when the forward ones agree)
            if not same_f:
                assert _count_differs_on_tolerance_edge(ch, 0, got[0], itf), (part, c, got, (itf, itb), ch.trace(0))
            else:
                assert _count_differs_on_tolerance_edge(ch, 1, got[1], itb), (part, c, got, (itf, itb), ch.trace(1))
            edge.append(c)
            ctol = 1e-7
        assert np.abs(q1[c] - qo).max() <= ctol * max(1.0, np.abs(qo).max()), (part, c)
        assert np.abs(p1[c] - po).max() <= ctol * max(1.0, np.abs(po).max()), (part, c)
        n_ok += st == 0
    assert len(edge) <= max(1, B // 30), edge  # tolerance-edge chains are rare
    return res, n_ok


def _spread_chains_by_stepping(ctx, case, part, rng, n_pre=2):
    """B DIFFERENT on-manifold states for data sets where they cannot be written down (noiseless observations: every
    chain has to reproduce the same observed path exactly): start every chain from chain 0's on-manifold state with its
    own momentum and step size and move it n_pre leapfrog steps with the library.  The states read back are then the
    common INPUT of the library and of the oracle for the step that is compared."""
    B = case["B"]
    ctx.set_state(np.repeat(case["q"][:1], B, 0), rng.standard_normal((B, ctx.Q)), np.repeat(case["x_obs"][:1], B, 0), part)
    ctx.project_onto_cotangent_space()
    pre = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.03 + 0.05 * rng.random(B))
    for _ in range(n_pre):
        r = ctx.leapfrog_step(pre)
        assert (r["status"] == 0).mean() > 0.9
    q, _, xo, _ = ctx.get_state()
    assert np.abs(q - q[:1]).max(1).min(initial=np.inf, where=np.arange(B) > 0) > 1e-3  # the chains have moved apart
    return q, xo


@pytest.mark.parametrize("name,model,T,S,R,noisy,obs_interval", [
    ("configs[2] FHN noiseless S=400", "fhn", 100, 400, 5, False, None),
    ("configs[4] shape FHN noisy S=800", "fhn", 100, 800, 5, True, None),
    ("configs[3] shape SIR S=200 one block", "sir", 14, 200, 14, True, 0.25)])
def test_full_size_distinct_chains_other_baseline_shapes(name, model, T, S, R, noisy, obs_interval):
    """The full-size check of configs[1] (test_full_size_distinct_chains_with_masked_subset) for the other BASELINE
    shapes: 72 DISTINCT chains (full and partial wavefronts of every kernel; the SIR single-block layout runs the
    time-parallel scan with chains that need different numbers of sweeps), a subset masked out, two chains whose
    retraction cannot converge -- every chain against the C oracle, in every partition."""
    B = 72
    rng = np.random.default_rng(72)
    if noisy:
        case = _distinct_on_manifold_chains(model, T, S, R, B, seed=73, obs_interval=obs_interval)
    else:
        case = make_case(model, T, S, R, noisy, B=B, seed=73, obs_interval=obs_interval)
    ctx = make_ctx(case)
    inactive, failing = [5, 17, 33, 64, 71], [7, 40]
    for part in range(ctx.num_partition):
        if noisy:
            q, xo = case["q"], case["x_obs"]
        else:
            q, xo = _spread_chains_by_stepping(ctx, case, part, rng)
        ctx.set_state(q, rng.standard_normal((B, ctx.Q)), xo, part)
        assert np.abs(ctx.constr()).max() < 1e-8
        ctx.project_onto_cotangent_space()
        _, p0, _, _ = ctx.get_state()
        h = 0.02 if model == "sir" else 0.04
        dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.5 * h + h * rng.random(B))
        dts[failing] = 5.0
        act = np.ones(B, dtype=np.int32)
        act[inactive] = 0
        res, n_ok = _step_every_chain_against_oracle(ctx, case["osys"], q, p0, xo, part, dts, act, max_iters=12,
                                                     edge_ok=model == "sir")  # (the only time-parallel-scan layout here)
        assert (res["status"][failing] > 0).all()
        assert n_ok >= B - len(inactive) - len(failing) - 4, (name, part, n_ok)
    ctx.close()


_MFMA_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from helpers import check_ops_against_oracle
from oracle import c_oracle
from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
B = 6
wl = SirWorkload(B, num_steps_per_obs=200)
ctx = wl.ctx
assert ctx.RM == 16 and ctx.K == [1]
q, _, xo, _ = ctx.get_state()
osys = c_oracle.OracleSystem("sir", 1.0, 200, 14, wl.y[:, 0], sigma=1.0)
case = dict(osys=osys, q=q, x_obs=xo, B=B, rng=np.random.default_rng(5))
worst = check_ops_against_oracle(ctx, case, tol=1e-9)
ctx.set_state(q, case["rng"].standard_normal((B, ctx.Q)), xo, 0)
ctx.project_onto_cotangent_space()
_, p0, _, _ = ctx.get_state()
dts = np.array([0.05, -0.05, 0.1, -0.1, 0.02, 0.2])
d0 = ctx.diagnostics()
res = ctx.leapfrog_step(dts)
d1 = ctx.diagnostics()
q1, p1, _, _ = ctx.get_state()
n_ok = 0
for c in range(B):
    ch = c_oracle.OracleChain(osys)
    ch.set(q[c], p0[c], xo[c], 0)
    st, itf, itb, _ = ch.step(dts[c])
    qo, po, _, _ = ch.get()
    assert res["status"][c] == st and res["iters_fwd"][c] == itf and (st != 0 or res["iters_bwd"][c] == itb), c
    assert np.abs(q1[c] - qo).max() <= 1e-9 * max(1.0, np.abs(qo).max()), c
    assert np.abs(p1[c] - po).max() <= 1e-9 * max(1.0, np.abs(po).max()), c
    n_ok += st == 0
assert n_ok >= 4
iters = int(res["iters_fwd"].max() + res["iters_bwd"].max())
print("MFMA_LAUNCHES", d1["gram_mfma_launches"], d1["gram_mfma_launches"] - d0["gram_mfma_launches"],
      "VALU_LAUNCHES", d1["gram_valu_launches"], "ITERS", iters, "CHOL_D", worst["chol_D"])
"""


def test_fp64_mfma_gram_kernel_boarding_school_sir():
    """BASELINE.json configs[4] names an fp64-MFMA J J^T Gram build.  The kernel (k_gram_rows_mfma,
    v_mfma_f64_16x16x4_f64 over the stored rows of 16-row blocks) is optional -- the vector-FMA kernel is faster on this
    part -- so it is run here in a child process with CHMC_GRAM_MFMA=1 (the
    stored-rows Newton sweep, the only path that forms a Gram block from rows): every per-op entry point (chol_D is the
    factor of the MFMA-built Gram) and one leapfrog step of the boarding-school SIR chains against the C oracle, and the
    library's own launch counters must show that the MFMA kernel, not the vector kernel, did the work."""
    import subprocess
    script = _MFMA_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    env = {**os.environ, "CHMC_GRAM_MFMA": "1"}
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("MFMA_LAUNCHES")][-1].split()
    total, in_step, valu, iters = int(line[1]), int(line[2]), int(line[4]), int(line[6])
    assert valu == 0, line            # the vector-FMA Gram kernel never ran
    assert in_step >= iters + 1, line  # one MFMA Gram per Newton iteration of the step + the state evaluation
    assert total > in_step, line      # ... and the state evaluations of the per-op checks


def test_sir_boarding_school_s200_adam_init_against_oracle():
    """BASELINE.json configs[3] as scripts/sir_model_chmc_experiment.py sets it up: the boarding-school counts, S = 200,
    ONE sub-sequence of R = 14 observations (16-row kernels), sigma_y = 1, initial states by the Adam-based finder of the
    noisy system (sde/mici_extensions.py:1679-1801) run on the library's operators.  The found states must lie on the
    manifold; every per-op entry point and one leapfrog step are checked against the C oracle AT those states."""
    from oracle import c_oracle
    from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
    B = 6
    wl = SirWorkload(B, num_steps_per_obs=200)
    ctx = wl.ctx
    assert ctx.Q == 8419 and ctx.C == [14] and ctx.K == [1] and ctx.RM == 16
    assert np.abs(ctx.constr()).max() < 1e-9
    q, _, xo, _ = ctx.get_state()
    assert (np.mean(q[:, -14:] ** 2, 1) < 1.0).all()  # the finder's acceptance threshold on the mean squared residual
    osys = c_oracle.OracleSystem("sir", 1.0, 200, 14, wl.y[:, 0], sigma=1.0)
    case = dict(osys=osys, q=q, x_obs=xo, B=B, rng=np.random.default_rng(5))
    check_ops_against_oracle(ctx, case, tol=1e-9)
    # one leapfrog step of every chain from ITS OWN state
    ctx.set_state(q, case["rng"].standard_normal((B, ctx.Q)), xo, 0)
    ctx.project_onto_cotangent_space()
    _, p0, _, _ = ctx.get_state()
    dts = np.array([0.05, -0.05, 0.1, -0.1, 0.02, 0.2])
    res = ctx.leapfrog_step(dts)
    q1, p1, _, _ = ctx.get_state()
    for c in range(B):
        ch = c_oracle.OracleChain(osys)
        ch.set(q[c], p0[c], xo[c], 0)
        st, itf, itb, _ = ch.step(dts[c])
        qo, po, _, _ = ch.get()
        assert res["status"][c] == st and res["iters_fwd"][c] == itf and (st != 0 or res["iters_bwd"][c] == itb)
        assert np.abs(q1[c] - qo).max() <= 1e-9 * max(1.0, np.abs(qo).max())
        assert np.abs(p1[c] - po).max() <= 1e-9 * max(1.0, np.abs(po).max())
    assert (res["status"] == 0).sum() >= 4
    ctx.close()


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,newton", [
    ("fhn", 12, 16, 5, True, False, True), ("fhn", 7, 8, 3, False, True, True), ("fhn", 12, 10, 5, True, False, False),
    ("sir", 14, 8, 14, True, False, True), ("sir", 6, 16, 2, True, False, True)])
def test_half_batches_on_two_streams_equal_one_batch(monkeypatch, model, T, S, R, noisy, gaussian, newton):
    """chmc_leapfrog_step as two overlapped half-batches (two streams, staggered forward scans, lock-step Newton loops)
    against the same step as one batch: bitwise equal, with masked and failing chains in both halves (71 chains: odd
    split 35 / 36, partial wavefronts)."""
    from helpers import halves_vs_single_batch
    case = make_case(model, T, S, R, noisy, B=71, seed=81, gaussian=gaussian)
    for part in range(2 if R and R < T else 1):
        halves_vs_single_batch(case, monkeypatch, part=part, newton=newton, masked=(3, 36, 70), failing=(5, 40))


def test_half_batches_full_size(monkeypatch):
    """The same at BASELINE.json configs[1]'s size (Q = 80106), 130 chains."""
    from helpers import halves_vs_single_batch
    case = make_case("fhn", 100, 400, 5, True, B=130, seed=82)
    halves_vs_single_batch(case, monkeypatch, part=1, n_steps=2, masked=(0, 64, 65, 129), failing=(7, 100))


def test_half_batches_with_several_check_workgroups_per_half(monkeypatch):
    """600 chains = 300 per half-batch: the convergence check of a half is then TWO workgroups whose last one publishes
    the round's count of iterating chains to the host (k_run_publish).  The two halves' checks run on different streams
    and may overlap, so each poll slot has its own ticket word; with a shared one a workgroup of the wrong launch
    publishes a partial count, the host stops a half early and unconverged chains pass as status 0.  Bitwise equal to
    the one-batch step, chains with different iteration counts and failing chains in both halves."""
    from helpers import halves_vs_single_batch
    case = make_case("fhn", 12, 16, 5, True, B=600, seed=83)
    for part in (0, 1):
        a = halves_vs_single_batch(case, monkeypatch, part=part, n_steps=3, masked=(3, 299, 300, 599),
                                   failing=(5, 256, 301, 580))
        counts = a[2][-1]["iters_fwd"]
        assert len(set(counts[a[2][-1]["status"] == 0].tolist())) >= 2  # the rounds are not all alike


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,newton,n_inner", [
    ("fhn", 12, 16, 5, True, False, True, 2), ("fhn", 7, 8, 3, False, True, True, 3), ("fhn", 12, 10, 5, True, False, False, 2),
    ("sir", 14, 8, 14, True, False, True, 2), ("sir", 6, 16, 2, True, False, True, 2), ("fhn_nb", 7, 8, 3, False, True, True, 2)])
def test_inner_h2_flow_steps(model, T, S, R, noisy, gaussian, newton, n_inner):
    """n_inner_step > 1 on the fused device path (mici _step_b; scripts/utils.py:131-136 --num-inner-h2-step)."""
    case = make_case(model, T, S, R, noisy, B=5, seed=91, gaussian=gaussian)
    ctx = make_ctx(case)
    dts = np.array([0.06, -0.06, 0.1, 0.03, -0.08])
    for part in range(ctx.num_partition):
        check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=2, n_inner=n_inner, part=part)
    check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=1, n_inner=n_inner, project=False)
    ctx.close()


def test_failure_in_a_later_inner_step_restores_the_start_state():
    from helpers import check_late_inner_failure
    found = 0
    for seed in range(93, 101):  # (which chain has the larger round-off in its last inner step depends on the arithmetic)
        case = make_case("fhn", 12, 16, 5, True, B=6, seed=seed)
        ctx = make_ctx(case)
        found += check_late_inner_failure(ctx, case, np.array([0.08, -0.08, 0.1, 0.05, -0.1, 0.07])) is not None
        ctx.close()
        if found >= 2:
            break
    assert found >= 1


def test_inner_steps_with_half_batches(monkeypatch):
    """n_inner_step = 2 with the step run as two half-batches: bitwise equal to the one-batch run."""
    case = make_case("fhn", 12, 16, 5, True, B=9, seed=94)
    out = []
    for halves in ("1", "2"):
        monkeypatch.setenv("CHMC_HALVES", halves)
        ctx = make_ctx(case)
        ctx.set_state(np.repeat(case["q"][:1], 9, 0), np.random.default_rng(3).standard_normal((9, ctx.Q)),
                      np.repeat(case["x_obs"][:1], 9, 0), 0)
        dts = np.linspace(-0.1, 0.1, 9)
        dts[4] = 5.0
        r = [ctx.leapfrog_step(dts, n_inner_step=2, max_iters=12) for _ in range(2)]
        out.append((ctx.get_state()[:2], r))
        ctx.close()
    (s1, r1), (s2, r2) = out
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    for a, b in zip(r1, r2):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


# variable observation noise: sigma = generate_σ_y(u) = exp(u[dim_z]), dim_u = dim_z + 1
# (sde/mici_extensions.py:353-358, 559-569, 601-608; scripts/sir_model_chmc_experiment.py:44,58,77)
VS_CASES = [("sir", 5, 6, None, False), ("sir", 6, 8, 2, False), ("sir", 6, 16, 2, False), ("fhn", 6, 8, 2, False),
            ("fhn", 7, 5, 3, True), ("sir", 14, 8, 14, False)]


@pytest.mark.parametrize("model,T,S,R,gaussian", VS_CASES)
def test_variable_observation_noise(model, T, S, R, gaussian):
    case = make_case(model, T, S, R, True, B=23, seed=11, gaussian=gaussian, var_sigma=True)
    ctx = make_ctx(case)
    assert ctx.U == 5 and ctx.Q == case["q"].shape[1]
    check_ops_against_oracle(ctx, case)
    ctx.close()
    case = make_case(model, T, S, R, True, B=5, seed=12, gaussian=gaussian, var_sigma=True)
    ctx = make_ctx(case)
    dts = np.array([0.05, -0.05, 0.1, 0.02, -0.08])
    for newton in ((True,) if R == 14 else (True, False)):  # (14-row quasi-Newton counts sit on the tolerance's edge)
        for part in range(ctx.num_partition):
            check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=2, part=part)
    check_steps_against_oracle(ctx, case, dts, n_steps=1, project=False)
    check_steps_against_oracle(ctx, case, dts, n_steps=1, n_inner=2)
    ctx.close()


@pytest.mark.parametrize("model,T,S,R", [("sir", 6, 8, 2), ("fhn", 6, 8, 2)])
def test_variable_observation_noise_block_metric(model, T, S, R):
    """Odd dim_u = 5 with M = blockdiag(M_0, I)."""
    from helpers import check_block_metric_against_oracle
    case = make_case(model, T, S, R, True, B=4, seed=41, var_sigma=True)
    ctx = make_ctx(case)
    check_block_metric_against_oracle(ctx, case, True, np.array([0.05, -0.05, 0.08, 0.02]))
    ctx.close()


def test_sir_variable_noise_boarding_school_s200():
    """The reference's SIR sweep with `--observation-noise-std -1` (scripts/run_sir_model_experiments.sh:8,
    sir_model_chmc_experiment.py:44,58,77): sigma = generate_σ_y(u), dim_u = 5, S = 200, one block of 14 rows; Adam-based
    initial states with the sigma-dependent objective, then ops and one step against the C oracle at those states."""
    from oracle import c_oracle
    from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
    B = 6
    wl = SirWorkload(B, num_steps_per_obs=200, sigma="variable")
    ctx = wl.ctx
    assert ctx.U == 5 and ctx.Q == 8420 and ctx.RM == 16
    assert np.abs(ctx.constr()).max() < 1e-9
    q, _, xo, _ = ctx.get_state()
    osys = c_oracle.OracleSystem("sir", 1.0, 200, 14, wl.y[:, 0], sigma="variable")
    case = dict(osys=osys, q=q, x_obs=xo, B=B, rng=np.random.default_rng(5))
    check_ops_against_oracle(ctx, case, tol=1e-9)
    ctx.set_state(q, case["rng"].standard_normal((B, ctx.Q)), xo, 0)
    ctx.project_onto_cotangent_space()
    _, p0, _, _ = ctx.get_state()
    dts = np.array([0.05, -0.05, 0.1, -0.1, 0.02, 0.2])
    res = ctx.leapfrog_step(dts)
    q1, p1, _, _ = ctx.get_state()
    for c in range(B):
        ch = c_oracle.OracleChain(osys)
        ch.set(q[c], p0[c], xo[c], 0)
        st, itf, itb, _ = ch.step(dts[c])
        qo, po, _, _ = ch.get()
        assert res["status"][c] == st and res["iters_fwd"][c] == itf and (st != 0 or res["iters_bwd"][c] == itb)
        assert np.abs(q1[c] - qo).max() <= 1e-9 * max(1.0, np.abs(qo).max())
        assert np.abs(p1[c] - po).max() <= 1e-9 * max(1.0, np.abs(po).max())
    assert (res["status"] == 0).sum() >= 4
    ctx.close()


_PATH_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from helpers import make_case, make_ctx
B = 6
case = make_case({model!r}, {T}, {S}, {R}, True, B=B, seed=77{extra})
ctx = make_ctx(case)
rng = np.random.default_rng(5)
out = {{}}
for part in range(ctx.num_partition):
    ctx.set_state(np.repeat(case["q"][:1], B, 0), rng.standard_normal((B, ctx.Q)), np.repeat(case["x_obs"][:1], B, 0), part)
    ctx.project_onto_cotangent_space()
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.02 + 0.01 * np.arange(B))
    res = [ctx.leapfrog_step(dts) for _ in range(2)]
    q, p, xo, _ = ctx.get_state()
    ctx.switch_partition()
    q2, p2, xo2, _ = ctx.get_state()
    out.update({{f"q{{part}}": q, f"p{{part}}": p, f"xo{{part}}": xo2, f"st{{part}}": np.stack([r["status"] for r in res]),
                f"it{{part}}": np.stack([r["iters_fwd"] + r["iters_bwd"] for r in res]), f"h{{part}}": ctx.hamiltonian()}})
np.savez({out!r}, **out)
"""


@pytest.mark.parametrize("model,T,S,R,extra", [("fhn", 100, 400, 5, ""), ("sir", 14, 200, 14, ", obs_interval=0.25")])
def test_compact_row_kernels_agree_with_the_stored_row_kernels_full_size(tmp_path, model, T, S, R, extra):
    """The library's two kernel families for the same step at the full BASELINE sizes: the default path (compact rows PB /
    LF: two-phase Newton sweep + fused factor / solve kernel, lean state and grad-log-det sweeps, time-parallel x_obs at
    the partition switch) against the stored-rows kernels of round 1 (CHMC_COMPACT_ROWS=0, the only family switch left),
    each in its own process (the switch is read once).
    Two leapfrog steps per partition and a partition switch: positions, momenta, x_obs to 1e-9 relative, statuses and
    Newton iteration counts equal."""
    import subprocess
    outs = []
    for name, env in (("compact", {}), ("stored", {"CHMC_COMPACT_ROWS": "0"})):
        out = str(tmp_path / f"{name}.npz")
        script = _PATH_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"), model=model, T=T, S=S, R=R, extra=extra, out=out)
        r = subprocess.run([sys.executable, "-c", script], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(out))
    a, b = outs
    for k in a.files:
        if k.startswith(("st", "it")):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        else:
            scale = max(np.abs(b[k]).max(), 1.0)
            assert np.abs(a[k] - b[k]).max() <= 1e-9 * scale, (k, np.abs(a[k] - b[k]).max(), scale)


def test_time_parallel_scan_against_the_sequential_scan(monkeypatch):
    """The SIR single-block layout three ways: (a) the default -- the whole leapfrog step of a chain in ONE launch by its own
    workgroup (k_traj_chain: retractions with time-parallel scans of 512 segments, every chain iterating as long as IT
    needs, state evaluation and momentum projection in the same workgroup); (b) CHMC_RETRACT_KERNEL=0: the lock-step rounds of round 3 with the
    time-parallel scan (carried-over sweeps, chain mask 2); (c) also CHMC_PAR_SCAN=0: lock-step rounds with the sequential
    lane-per-block scan.  Statuses equal, Newton iteration counts equal (up to tolerance-edge counts: the scans agree to
    1e-12, not bitwise), positions to 1e-9, over 64 distinct chains x 6 steps with two chains whose retraction diverges."""
    B = 64
    case = _distinct_on_manifold_chains("sir", 14, 200, 14, B, seed=74, obs_interval=0.25)
    rng = np.random.default_rng(12)
    p = rng.standard_normal(case["q"].shape)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.01 + 0.03 * rng.random(B))
    dts[[9, 33]] = 5.0
    out = []
    for retract, par in (("1", "1"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("CHMC_RETRACT_KERNEL", retract)
        monkeypatch.setenv("CHMC_PAR_SCAN", par)
        ctx = make_ctx(case)
        ctx.set_state(case["q"], p, case["x_obs"], 0)
        ctx.project_onto_cotangent_space()
        res = [ctx.leapfrog_step(dts, max_iters=15) for _ in range(6)]
        q1, p1, _, _ = ctx.get_state()
        d = ctx.diagnostics()
        out.append((res, q1, p1, int(d["par_scan"][1:48].sum()), d["retract_kernel_launches"] + d["traj_kernel_launches"]))
        ctx.close()
    (ra, qa, pa, na, ka), (rl, ql, pl, nl, kl), (rb, qb, pb, nb, kb) = out
    assert ka == 6 and kl == 0 and kb == 0  # one launch per step (k_traj_chain), in the first context only
    assert na > 0 and nl > 0 and nb == 0     # time-parallel sweeps ran in the first two contexts only
    # (a) = (b) BIT FOR BIT: the batched path of this layout runs the per-chain kernels' arithmetic (512 segments per chain,
    # workgroup-parallel combine, 16-lane factorisations), so the execution model -- one workgroup per chain or batched
    # launches, chosen by the number of chains per CU -- never changes a result
    for x, y in zip(ra, rl):
        for k in x:
            np.testing.assert_array_equal(x[k], y[k], err_msg=k)
    np.testing.assert_array_equal(qa, ql)
    np.testing.assert_array_equal(pa, pl)
    # against the sequential scan: statuses equal, counts equal up to tolerance-edge cases, positions to 1e-9
    differ = 0
    for x, y in zip(ra, rb):
        np.testing.assert_array_equal(x["status"], y["status"])
        differ += int((x["iters_fwd"] != y["iters_fwd"]).sum() + (x["iters_bwd"] != y["iters_bwd"]).sum())
    assert differ <= 2  # (a count may differ on the edge of a tolerance)
    if differ == 0:
        assert np.abs(qa - qb).max() <= 1e-9 * max(1.0, np.abs(qb).max())
    assert (ra[0]["status"][[9, 33]] > 0).all()


@pytest.mark.parametrize("T,S,B,var_sigma", [(14, 8, 37, False), (12, 16, 9, True), (14, 200, 40, False)])
def test_per_chain_kernels_equal_the_batched_path_bitwise(monkeypatch, T, S, B, var_sigma):
    """Layouts with one 16-row block per chain (SIR, R >= T): k_traj_chain (default: one workgroup per chain, whole steps)
    against the batched lock-step path of the same layouts (CHMC_RETRACT_KERNEL=0: k_fwd_par<8>, k_newton_ivl,
    k_newton_comb_wg, KUpdatePB, KCheck, ... as separate launches): positions, momenta, statuses, iteration counts,
    reverse-check distances and Hamiltonians bitwise equal over 4 steps, with masked and failing chains, short and long
    blocks, fixed and variable observation noise."""
    case = make_case("sir", T, S, T, True, B=B, seed=61, obs_interval=0.25, var_sigma=var_sigma)
    rng = np.random.default_rng(8)
    p = rng.standard_normal(case["q"].shape)
    qq, xx = np.repeat(case["q"][:1], B, 0), np.repeat(case["x_obs"][:1], B, 0)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.01 + 0.03 * rng.random(B))
    dts[[2, B - 2]] = 5.0
    act = np.ones(B, dtype=np.int32)
    act[[1, B - 1]] = 0
    out = []
    for retract in ("1", "0", "2"):
        monkeypatch.setenv("CHMC_RETRACT_KERNEL", retract)
        ctx = make_ctx(case)
        assert ctx.RM == 16 and ctx.K == [1]
        ctx.set_state(qq, p, xx, 0)
        ctx.project_onto_cotangent_space()
        res = [ctx.leapfrog_step(dts, active=act if k == 0 else None, max_iters=15) for k in range(3)]
        res.append(ctx.leapfrog_steps(dts, 2, max_iters=15))
        q1, p1, _, _ = ctx.get_state()
        d = ctx.diagnostics()
        out.append((res, q1, p1, ctx.hamiltonian(), d["traj_kernel_launches"]))
        ctx.close()
    (ra, qa, pa, ha, ka), (rb, qb, pb, hb, kb) = out[:2]
    assert ka == 4 and kb == 0 and out[2][4] == 4
    for rb, qb, pb, hb, kb in out[1:]:  # (the batched path; the per-chain kernels with four wavefronts, two chains per CU)
        for x, y in zip(ra, rb):
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=k)
        np.testing.assert_array_equal(qa, qb)
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ha, hb)
    assert (ra[0]["status"][[1, B - 1]] == -1).all() and (ra[0]["status"][[2, B - 2]] > 0).all()
    assert (ra[-1]["n_done"] == 2).sum() >= B // 2


@pytest.mark.parametrize("B,shards", [(256, 2), (576, 3), (1280, 5)])
def test_results_do_not_depend_on_the_shard_size(B, shards):
    """SURVEY 4 (viii) / BASELINE configs[3] (1 024 SIR chains sharded over 4 GPUs): a chain's results must not depend on
    how many chains share its context.  Boarding-school chains (Adam-based initial states, S = 200, one 14-row block)
    stepped as ONE context and as `shards` contexts of B / shards chains with the default switches: positions, momenta,
    statuses, iteration counts and reverse-check distances after 3 steps agree BITWISE (every kernel choice that changes
    arithmetic follows from the layout; the scans use a fixed number of segments per chain; every reduction has a fixed
    order).  576 chains: the one context runs the per-chain kernels with FOUR wavefronts per chain (two chains per compute
    unit), its shards of 192 with eight; 1 280 chains: the one context runs the BATCHED path (more than four chains per compute
    unit), its shards of 256 the per-chain kernels -- the execution models give the same bits."""
    from manifold_mcmc_for_diffusions_amd.workload import SirWorkload
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    wl = SirWorkload(B, num_steps_per_obs=200)
    q0, _, xo, _ = wl.ctx.get_state()
    rng = np.random.default_rng(31)
    p0 = rng.standard_normal(q0.shape)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.1 + 0.2 * rng.random(B))

    def run(ctx, sl):
        ctx.set_state(q0[sl], p0[sl], xo[sl], 0)
        ctx.project_onto_cotangent_space()
        d0 = ctx.diagnostics()["traj_kernel_launches"]
        res = [ctx.leapfrog_step(dts[sl], **wl.solver) for _ in range(3)]
        q1, p1, _, _ = ctx.get_state()
        return res, q1, p1, ctx.diagnostics()["traj_kernel_launches"] - d0

    whole = run(wl.ctx, slice(0, B))
    assert whole[3] == (3 if B <= 1024 else 0)  # (MI355X: 256 compute units; per-chain kernels up to four chains per CU)
    wl.ctx.close()
    n_ok, Bs = 0, B // shards
    for h in range(shards):
        sl = slice(h * Bs, (h + 1) * Bs)
        ctx = ChmcContext("sir", 1.0, 200, 14, wl.y[:, 0], sigma=1.0, num_chains=Bs)
        part = run(ctx, sl)
        ctx.close()
        assert part[3] == 3
        for ra, rb in zip(whole[0], part[0]):
            for k in ra:
                np.testing.assert_array_equal(ra[k][sl], rb[k], err_msg=f"{k} shard {h}")
        np.testing.assert_array_equal(whole[1][sl], part[1])
        np.testing.assert_array_equal(whole[2][sl], part[2])
        n_ok += int((part[0][-1]["status"] == 0).sum())
    assert n_ok >= B // 2  # (the comparison is of moving chains, not of failed steps that left their state alone)


def test_time_parallel_scan_absorbed_and_nan_trajectories(monkeypatch):
    """k_fwd_par on trajectories with absorbed compartments (SIR clips log S and log I at -500 and keeps a clipped
    component where it is, sde/example_models/sir.py:54-70) and with NaNs, from guesses that are absorbed where the
    trajectory is not and the other way round: the unconstrained target (one forward scan of T S = 2 800 steps per chain,
    conditioned_diffusion_neg_log_dens_and_grad, sde/mici_extensions.py:82-205) of healthy, wild, then healthy points
    again -- every scan is seeded with the previous call's trajectory -- with 1, 2 and 4 wavefronts per chain against the
    sequential scan: same NaN pattern, values and gradients to 1e-9; two chains per call against the autodiff oracle."""
    from oracle.py.neg_log_dens import neg_log_dens_and_grad
    B, T, S = 96, 14, 200
    case = make_case("sir", T, S, 14, True, B=B, seed=5, obs_interval=0.25)
    rng = np.random.default_rng(77)
    QH = None
    calls = None
    results = {}
    for mode in ("seq", "1", "2", "4"):
        monkeypatch.setenv("CHMC_PAR_SCAN", "0" if mode == "seq" else "1")
        monkeypatch.setenv("CHMC_PAR_WAVES", "1" if mode == "seq" else mode)
        ctx = make_ctx(case)
        if calls is None:
            QH = ctx.U + ctx.NV
            scale = np.repeat([0.5, 2.0, 4.0, 8.0], B // 4)[:, None]
            healthy = 0.5 * rng.standard_normal((B, QH))
            wild = scale * rng.standard_normal((B, QH))
            wild[:, :ctx.U] = scale * rng.standard_normal((B, ctx.U)) * 2.0
            calls = [healthy, wild, healthy + 0.01 * rng.standard_normal((B, QH)), wild[::-1].copy(), healthy]
        results[mode] = [ctx.neg_log_dens_and_grad(q) for q in calls]
        if mode != "seq":
            assert int(ctx.diagnostics()["par_scan"][1:48].sum()) > 0
        ctx.close()
    n_nan = 0
    for i, q in enumerate(calls):
        v0, g0 = results["seq"][i]
        n_nan += int(np.isnan(v0).sum())
        for mode in ("1", "2", "4"):
            v, g = results[mode][i]
            np.testing.assert_array_equal(np.isnan(v), np.isnan(v0), err_msg=f"call {i} W={mode}")
            ok = ~np.isnan(v0)
            assert np.abs(v[ok] - v0[ok]).max() <= 1e-9 * np.maximum(1.0, np.abs(v0[ok])).max(), (i, mode)
            gs = np.maximum(1.0, np.nanmax(np.abs(g0[ok]), axis=1, keepdims=True))
            np.testing.assert_array_equal(np.isnan(g[ok]), np.isnan(g0[ok]), err_msg=f"call {i} W={mode}")
            assert np.nanmax(np.abs(g[ok] - g0[ok]) / gs) <= 1e-9, (i, mode)
        for c in (0, B - 1):
            vo, go = neg_log_dens_and_grad("sir", case["obs_interval"], S, case["y"], case["sigma"], q[c], False)
            if np.isnan(vo):
                assert np.isnan(v0[c])
            else:
                assert abs(v0[c] - vo) <= 1e-9 * max(1.0, abs(vo)), (i, c, v0[c], vo)
    # the wild calls must actually contain absorbed / overflowing trajectories, or the test shows nothing
    xs = results["seq"][1][0]
    assert np.isfinite(xs).sum() >= B // 4


@pytest.mark.parametrize("split", ["1", "4"])
@pytest.mark.parametrize("T,S,R,B", [(14, 200, 14, 3), (26, 24, 13, 5), (16, 72, 16, 2)])
def test_sixteen_row_blocks_row_split_settings(monkeypatch, split, T, S, R, B):
    """16-row blocks (SIR): the two state evaluations (CHMC_ROW_SPLIT=1: the stored-rows sweeps k_rev_wave_ldsrows +
    k_gram_rows + k_gld_fwd_wave + k_gld_bwd_wave_ldsrows that large batches keep; otherwise, the default for up to 1 024
    blocks: the interval-parallel sweep on the compact rows k_newton_ivl / k_newton_comb<STATE> and the row-free grad-log-det
    sweeps with every observation interval on its own wavefront, k_gld_ivl_prologue / k_gld_fwd_ivl / k_gld_bwd_ivl<0, 1> /
    k_gld_ivl_finish) against the C oracle: every per-operator entry point (chol_gram_blocks :794-810,
    grad_log_det_sqrt_gram :1143-1146, jacob_constr_blocks :704-763 -- the rows are rebuilt from the compact form on demand)
    and two leapfrog steps; single-block layout at full size, two blocks per chain (13 observation rows + 3 state rows, then
    13 rows) and a full 16-row block."""
    monkeypatch.setenv("CHMC_ROW_SPLIT", split)
    case = make_case("sir", T, S, R, True, B=B, seed=23 + T, obs_interval=0.25)
    ctx = make_ctx(case)
    assert ctx.RM == 16
    # (the two-block prior draw has a poorly conditioned Gram matrix: inverse-Gram products agree to 1e-8 in every setting)
    check_ops_against_oracle(ctx, case, tol=1e-7 if T == 26 else 1e-9)
    check_steps_against_oracle(ctx, case, np.where(np.arange(B) % 2 == 0, 0.02, -0.02), n_steps=2)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("noisy,gaussian", [(True, False), (False, False), (True, True)])
def test_momentum_correction_inside_the_jp_pass_is_bitwise_the_separate_pass(monkeypatch, noisy, gaussian):
    """The momentum correction p -= dh2_flow_mom_dmom @ (mu / dt) and pg <- dh1_dpos of a step (mici _step_b,
    sde/mici_extensions.py:1233-1238) ride in the J p pass over the compact rows (k_jw_pb<.., FIX>; the u, v_0 and
    observation-noise columns by KMomFixEdges), and the reverse flow of the reversibility check rides in the J^T lambda pass
    (KUpdatePB<.., 3> with flow_rev) -- against CHMC_STEP_FUSIONS=0, the separate passes KMomFixInitPg and KFlow: positions,
    momenta, statuses, counts and Hamiltonians bitwise equal over 3 steps in both partitions, with masked and failing chains."""
    B = 37
    case = make_case("fhn", 10, 16, 5, noisy, B=B, seed=44, gaussian=gaussian)
    rng = np.random.default_rng(4)
    qq, xx = np.repeat(case["q"][:1], B, 0), np.repeat(case["x_obs"][:1], B, 0)
    p = rng.standard_normal(qq.shape)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.01 + 0.04 * rng.random(B))
    dts[5] = 5.0
    act = np.ones(B, dtype=np.int32)
    act[[2, B - 1]] = 0
    for part in (0, 1):
        out = []
        for off in (None, "1"):
            if off:
                monkeypatch.setenv("CHMC_STEP_FUSIONS", "0")
            else:
                monkeypatch.delenv("CHMC_STEP_FUSIONS", raising=False)
            ctx = make_ctx(case)
            ctx.set_state(qq, p, xx, part)
            ctx.project_onto_cotangent_space()
            res = [ctx.leapfrog_step(dts, active=act if k == 0 else None, max_iters=15) for k in range(3)]
            res.append(ctx.leapfrog_steps(np.abs(dts), 3, active=act, max_iters=15))  # (with the fused kicks of a trajectory)
            q1, p1, _, _ = ctx.get_state()
            out.append((res, q1, p1, ctx.hamiltonian()))
            ctx.close()
        (ra, qa, pa, ha), (rb, qb, pb, hb) = out
        for x, y in zip(ra, rb):
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=k)
        np.testing.assert_array_equal(qa, qb)
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ha, hb)
        assert ra[0]["status"][5] > 0 and (ra[0]["status"][[2, B - 1]] == -1).all() and (ra[0]["status"] == 0).sum() > B // 2
        assert (ra[2]["status"] == 0).sum() > B // 2 and (ra[3]["n_done"] == 3).sum() > B // 2
