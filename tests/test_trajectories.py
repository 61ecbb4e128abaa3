"""chmc_leapfrog_steps (whole trajectories in one library call: the loop an integration transition runs around
integrator.step, scripts/utils.py:284-301) against chmc_leapfrog_step applied repeatedly with the trajectory semantics
spelled out on the host (a chain stops at its first failed step) and against the C oracle.
CPU: the TEST-ONLY emulation build (the call's bookkeeping); `-m gpu`: the HIP library (tests/test_hip_parity.py holds
the full-size single-step cases)."""
import numpy as np
import pytest
from helpers import make_case, make_ctx
from test_emu_logic import emu_lib  # noqa: F401

import os

SOLVER = dict(max_iters=12)


def lockstep_trajectories(ctx, dts, n_steps, active=None, **kw):
    """The same trajectories with one batched chmc_leapfrog_step per step and the bookkeeping on the host."""
    B = ctx.B
    n_steps = np.broadcast_to(np.asarray(n_steps, dtype=np.int64), (B,)).copy()
    act = np.ones(B, dtype=np.int32) if active is None else np.asarray(active, dtype=np.int32).copy()
    left = np.where(act == 1, n_steps, 0)
    out = dict(n_done=np.zeros(B, np.int32), status=np.where(act == 1, 0, -1).astype(np.int32),
               iters_fwd=np.zeros(B, np.int32), iters_bwd=np.zeros(B, np.int32), rev_err=np.zeros(B))
    while (left > 0).any():
        run = (left > 0).astype(np.int32)
        r = ctx.leapfrog_step(dts, active=run, **kw)
        on = run == 1
        out["iters_fwd"][on] += r["iters_fwd"][on]
        out["iters_bwd"][on] += r["iters_bwd"][on]
        reached = on & (r["status"] != 1) & (r["status"] != 2)
        out["rev_err"][reached] = r["rev_err"][reached]
        ok = on & (r["status"] == 0)
        bad = on & (r["status"] > 0)
        out["n_done"][ok] += 1
        left[ok] -= 1
        out["status"][bad] = r["status"][bad]
        left[bad] = 0
    return out


def run_both(case, dts, n_steps, active=None, part=0, newton=True, seed=5, **kw):
    """The library call and the host loop from identical states; returns (library result, host-loop result) with states."""
    B = case["B"]
    rng = np.random.default_rng(seed)
    qq = np.repeat(case["q"][:1], B, 0)
    xx = np.repeat(case["x_obs"][:1], B, 0)
    p = rng.standard_normal(qq.shape)
    outs = []
    for engine in (True, False):
        ctx = make_ctx(case)
        ctx.set_state(qq, p, xx, part)
        ctx.project_onto_cotangent_space()
        if engine:
            r = ctx.leapfrog_steps(dts, n_steps, active=active, newton=newton, **kw)
        else:
            r = lockstep_trajectories(ctx, dts, n_steps, active=active, newton=newton, **kw)
        q1, p1, _, _ = ctx.get_state()
        outs.append((r, q1, p1, ctx.hamiltonian()))
        ctx.close()
    return outs


def assert_same(a, b, bitwise=True):
    (ra, qa, pa, ha), (rb, qb, pb, hb) = a, b
    for k in ("n_done", "status", "iters_fwd", "iters_bwd"):
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    if bitwise:
        assert np.array_equal(qa, qb) and np.array_equal(pa, pb) and np.array_equal(ha, hb)
        np.testing.assert_array_equal(ra["rev_err"], rb["rev_err"])
    else:
        np.testing.assert_allclose(qa, qb, rtol=0, atol=1e-9 * max(1.0, np.abs(qb).max()))
        np.testing.assert_allclose(pa, pb, rtol=0, atol=1e-9 * max(1.0, np.abs(pb).max()))


CASES = [
    ("fhn", 6, 4, 2, True, False), ("fhn", 7, 5, 3, False, False), ("fhn", 6, 4, 2, True, True),
    ("fhn", 12, 10, 5, False, True), ("sir", 5, 6, None, True, False), ("sir", 6, 8, 2, True, False),
    ("sir", 14, 6, 14, True, False),
]


@pytest.mark.parametrize("model,T,S,R,noisy,gaussian", CASES)
@pytest.mark.parametrize("newton", [True, False])
def test_trajectories_equal_repeated_steps(emu_lib, model, T, S, R, noisy, gaussian, newton):  # noqa: F811
    """Mixed step sizes (chains need different numbers of Newton iterations), per-chain trajectory lengths, inactive
    chains and a chain whose first step fails: bitwise the results of the host loop."""
    B = 7
    case = make_case(model, T, S, R, noisy, B=B, seed=21, gaussian=gaussian)
    dts = np.array([0.05, -0.05, 0.1, 0.02, 5.0, -0.08, 0.03])
    n_steps = np.array([3, 1, 4, 2, 3, 0, 4])
    active = np.array([1, 1, 1, 0, 1, 1, 1], dtype=np.int32)
    for part in range(2 if R and R < T else 1):
        a, b = run_both(case, dts, n_steps, active=active, part=part, newton=newton, **SOLVER)
        assert_same(a, b)
        r = a[0]
        assert r["status"][3] == -1 and r["n_done"][3] == 0 and r["n_done"][5] == 0 and r["status"][5] == 0
        assert r["status"][4] > 0 and r["n_done"][4] == 0
        assert (r["n_done"][[0, 1, 2, 6]] <= n_steps[[0, 1, 2, 6]]).all() and r["n_done"].sum() >= 6
        # every chain the same number of steps (an integration transition): the lock-step path fuses the closing A(dt/2) of
        # a step with the opening A(dt/2) + flow of the next (KKick2FlowPg); a chain that fails later keeps the kicked momentum
        dts2 = dts.copy()
        dts2[2] = 0.6  # (fails, if at all, after its first steps)
        a, b = run_both(case, dts2, 4, active=active, part=part, newton=newton, **SOLVER)
        assert_same(a, b)
        assert a[0]["status"][4] > 0 and (a[0]["n_done"][[0, 1, 6]] >= 1).all()


def test_unprojected_momenta(emu_lib):  # noqa: F811
    """Momenta set through set_state (not projected): the first step must take the reference's full path for its first
    half-kick (no tangent-momentum shortcut), in the library call as in the host loop."""
    case = make_case("fhn", 6, 4, 2, True, B=4, seed=22)
    dts = np.array([0.05, -0.05, 0.1, 0.02])
    B = 4
    ctx = make_ctx(case)
    rng = np.random.default_rng(3)
    qq, xx = np.repeat(case["q"][:1], B, 0), np.repeat(case["x_obs"][:1], B, 0)
    p = rng.standard_normal(qq.shape)
    ctx.set_state(qq, p, xx, 0)
    r1 = ctx.leapfrog_steps(dts, 2, **SOLVER)
    s1 = ctx.get_state()[:2]
    ctx.set_state(qq, p, xx, 0)
    r2 = lockstep_trajectories(ctx, dts, 2, **SOLVER)
    s2 = ctx.get_state()[:2]
    ctx.close()
    for k in ("n_done", "status", "iters_fwd", "iters_bwd"):
        np.testing.assert_array_equal(r1[k], r2[k])
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])


def test_trajectories_against_the_oracle(emu_lib):  # noqa: F811
    from oracle import c_oracle
    B = 5
    case = make_case("fhn", 12, 10, 5, True, B=B, seed=23)
    dts = np.array([0.06, -0.06, 0.1, 0.03, -0.08])
    n_steps = np.array([3, 2, 3, 1, 3])
    ctx = make_ctx(case)
    qq, xx = np.repeat(case["q"][:1], B, 0), np.repeat(case["x_obs"][:1], B, 0)
    ctx.set_state(qq, np.random.default_rng(1).standard_normal(qq.shape), xx, 0)
    ctx.project_onto_cotangent_space()
    _, p0, _, _ = ctx.get_state()
    r = ctx.leapfrog_steps(dts, n_steps)
    q1, p1, _, _ = ctx.get_state()
    for c in range(B):
        ch = c_oracle.OracleChain(case["osys"])
        ch.set(qq[c], p0[c], xx[c], 0)
        itf = itb = nd = 0
        st = 0
        for _ in range(n_steps[c]):
            st, f, b, _ = ch.step(dts[c])
            itf, itb = itf + f, itb + (b if st == 0 else 0)
            if st:
                break
            nd += 1
        qo, po, _, _ = ch.get()
        assert (r["n_done"][c], r["status"][c], r["iters_fwd"][c]) == (nd, st, itf), c
        assert st != 0 or r["iters_bwd"][c] == itb
        assert np.abs(q1[c] - qo).max() <= 1e-8 * max(1.0, np.abs(qo).max())
        assert np.abs(p1[c] - po).max() <= 1e-8 * max(1.0, np.abs(po).max())
    ctx.close()


# ---------------------------------------------------------------------------------------------------- HIP library
GPU_CASES = [
    ("fhn", 12, 16, 5, True, False, True), ("fhn", 12, 16, 5, False, False, True), ("fhn", 7, 8, 3, False, True, True),
    ("fhn", 12, 10, 5, True, False, False), ("sir", 6, 16, 2, True, False, True), ("sir", 14, 8, 14, True, False, True),
    ("fhn_nb", 7, 8, 3, False, True, True),
]


@pytest.mark.gpu
@pytest.mark.parametrize("model,T,S,R,noisy,gaussian,newton", GPU_CASES)
def test_trajectories_equal_repeated_steps_hip(model, T, S, R, noisy, gaussian, newton):
    """On the MI355X: 37 chains (partial wavefronts), mixed step sizes, per-chain trajectory lengths, inactive and failing
    chains -- bitwise the host loop's results."""
    from manifold_mcmc_for_diffusions_amd import _lib
    assert _lib.lib().chmc_backend() == b"hip:gfx950"
    B = 37
    case = make_case(model, T, S, R, noisy, B=B, seed=31, gaussian=gaussian)
    rng = np.random.default_rng(7)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.02 + 0.1 * rng.random(B))
    dts[[4, 20]] = 5.0
    n_steps = rng.integers(0, 5, B)
    active = np.ones(B, dtype=np.int32)
    active[[3, 36]] = 0
    for part in range(2 if R and R < T else 1):
        a, b = run_both(case, dts, n_steps, active=active, part=part, newton=newton, **SOLVER)
        assert_same(a, b)
        assert (a[0]["status"][[4, 20]] > 0).all() and a[0]["n_done"].sum() > B
        # the same number of steps for every chain: the fused kicks of the lock-step path (KKick2FlowPg)
        a, b = run_both(case, dts, 4, active=active, part=part, newton=newton, **SOLVER)
        assert_same(a, b)
        assert (a[0]["n_done"][a[0]["status"] == 0] == 4).all() and (a[0]["n_done"] == 4).sum() > B // 2


@pytest.mark.gpu
def test_trajectories_full_size_distinct_chains_hip():
    """BASELINE.json configs[1] at full size, 72 distinct chains, trajectories of 3 steps: the library call against the host
    loop (bitwise) and its first step against the C oracle chain by chain."""
    from oracle import c_oracle
    from test_hip_parity import _distinct_on_manifold_chains
    B = 72
    case = _distinct_on_manifold_chains("fhn", 100, 400, 5, B, seed=71)
    rng = np.random.default_rng(9)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.02 + 0.06 * rng.random(B))
    dts[[7, 40]] = 5.0
    active = np.ones(B, dtype=np.int32)
    active[[5, 64]] = 0
    p = rng.standard_normal(case["q"].shape)
    outs = []
    for engine in (True, False):
        ctx = make_ctx(case)
        ctx.set_state(case["q"], p, case["x_obs"], 1)
        ctx.project_onto_cotangent_space()
        p0 = ctx.get_state()[1]
        r = (ctx.leapfrog_steps(dts, 3, active=active, **SOLVER) if engine
             else lockstep_trajectories(ctx, dts, 3, active=active, **SOLVER))
        q1, p1, _, _ = ctx.get_state()
        outs.append((r, q1, p1, ctx.hamiltonian()))
        ctx.close()
    assert_same(*outs)
    r = outs[0][0]
    assert (r["status"][[7, 40]] > 0).all() and (r["n_done"][active == 1] >= 0).all() and (r["n_done"] == 3).sum() >= 60
    # one-step trajectories against the oracle (the multi-step ones are covered through the lock-step equality above)
    ctx = make_ctx(case)
    ctx.set_state(case["q"], p, case["x_obs"], 1)
    ctx.project_onto_cotangent_space()
    r1 = ctx.leapfrog_steps(dts, 1, active=active, **SOLVER)
    q1, p1, _, _ = ctx.get_state()
    for c in range(0, B, 3):
        if not active[c]:
            continue
        ch = c_oracle.OracleChain(case["osys"])
        ch.set(case["q"][c], p0[c], case["x_obs"][c], 1)
        st, itf, itb, _ = ch.step(dts[c], max_iters=12)
        qo, po, _, _ = ch.get()
        assert (r1["status"][c], r1["iters_fwd"][c]) == (st, itf) and (st != 0 or r1["iters_bwd"][c] == itb), c
        assert np.abs(q1[c] - qo).max() <= 1e-9 * max(1.0, np.abs(qo).max())
        assert np.abs(p1[c] - po).max() <= 1e-9 * max(1.0, np.abs(po).max())
    ctx.close()


@pytest.mark.gpu
def test_trajectories_sir_single_block_time_parallel_scan_hip():
    """BASELINE.json configs[3]'s shape (one 14-row block of 2 800 steps: per-chain retraction kernel, time-parallel scans,
    interval-parallel 16-row state evaluation): 72 distinct chains, trajectories of 4 steps against the host loop --
    statuses, step counts and iteration counts equal, positions to 1e-9."""
    from test_hip_parity import _distinct_on_manifold_chains
    B = 72
    case = _distinct_on_manifold_chains("sir", 14, 200, 14, B, seed=73, obs_interval=0.25)
    rng = np.random.default_rng(11)
    dts = np.where(np.arange(B) % 2 == 0, 1.0, -1.0) * (0.01 + 0.02 * rng.random(B))
    dts[[7, 40]] = 5.0
    p = rng.standard_normal(case["q"].shape)
    outs = []
    for engine in (True, False):
        ctx = make_ctx(case)
        ctx.set_state(case["q"], p, case["x_obs"], 0)
        ctx.project_onto_cotangent_space()
        r = ctx.leapfrog_steps(dts, 4, **SOLVER) if engine else lockstep_trajectories(ctx, dts, 4, **SOLVER)
        q1, p1, _, _ = ctx.get_state()
        outs.append((r, q1, p1))
        ctx.close()
    (ra, qa, pa), (rb, qb, pb) = outs
    np.testing.assert_array_equal(ra["status"], rb["status"])
    np.testing.assert_array_equal(ra["n_done"], rb["n_done"])
    assert (ra["iters_fwd"] != rb["iters_fwd"]).sum() + (ra["iters_bwd"] != rb["iters_bwd"]).sum() <= 3
    same = (ra["iters_fwd"] == rb["iters_fwd"]) & (ra["iters_bwd"] == rb["iters_bwd"])
    assert np.abs(qa[same] - qb[same]).max() <= 1e-9 * max(1.0, np.abs(qb).max())
    assert np.abs(qa - qb).max() <= 1e-7 * max(1.0, np.abs(qb).max())
    assert (ra["status"][[7, 40]] > 0).all() and (ra["n_done"] == 4).sum() >= 60
