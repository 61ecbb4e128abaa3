"""The batched dynamic (no-U-turn, multinomial) transition of dynamic.py against a plain single-chain restatement
that integrates every sub-tree leaf by leaf and applies the no-U-turn criterion to the aligned binary spans directly
(the recursive definition), with the same keyed random draws -- through the TEST-ONLY emulation build (CPU) and, under
`-m gpu`, through the HIP library itself (chmc_tree_step / KTreeDecide / KTreeTurn on the device)."""
import numpy as np
from test_emu_logic import emu_lib  # noqa: F401


def _reference_transition(ctx1, q, p, xo, part, draw, eps, max_depth, max_delta_h, solver, W=None, extra=True):
    def vel(pp):  # dh_dmom = metric.inv @ mom (sde/mici_extensions.py:1204-1208)
        if W is None:
            return pp
        v = pp.copy()
        v[:W.shape[0]] = W @ pp[:W.shape[0]]
        return v

    def step(qv, pv, sgn):
        ctx1.set_state(qv[None], pv[None], xo[None], part)
        r = ctx1.leapfrog_step(np.array([sgn * eps]), **solver)
        if r["status"][0] != 0:
            return None
        q2, p2, _, _ = ctx1.get_state()
        return q2[0], p2[0], ctx1.hamiltonian()[0, 0]

    ctx1.set_state(q[None], p[None], xo[None], part)
    h0 = ctx1.hamiltonian()[0, 0]
    neg, pos, sum_mom, logw, prop = (q, p), (q, p), p.copy(), -h0, q
    n_step = 0
    for d in range(max_depth):
        fwd = draw(0, d, 0) < 0.5
        cur = pos if fwd else neg
        leaves, ok, sub_logw, sub_prop = [], True, -np.inf, None
        for k in range(1 << d):
            out = step(cur[0], cur[1], 1.0 if fwd else -1.0)
            if out is None or not (out[2] - h0 <= max_delta_h):
                ok = False
                break
            q2, p2, h = out
            n_step += 1
            leaves.append((q2, p2))
            cur = (q2, p2)
            new = np.logaddexp(sub_logw, -h)
            if draw(2, d, k) < np.exp(-h - new):
                sub_prop = q2
            sub_logw = new
            j = 1
            while (k + 1) % (1 << j) == 0 and (1 << j) <= (1 << d):  # aligned spans of 2^j leaves ending here
                a = k + 1 - (1 << j)
                span = sum(pp for _, pp in leaves[a:k + 1])
                if vel(leaves[a][1]) @ span < 0 or vel(leaves[k][1]) @ span < 0:
                    ok = False
                if extra and j >= 2:  # Mici's additional checks across the two halves of the span
                    m = a + (1 << (j - 1)) - 1
                    rho1 = sum(pp for _, pp in leaves[a:m + 1]) + leaves[m + 1][1]
                    rho2 = sum(pp for _, pp in leaves[m + 1:k + 1]) + leaves[m][1]
                    if vel(leaves[a][1]) @ rho1 < 0 or vel(leaves[m + 1][1]) @ rho1 < 0:
                        ok = False
                    if vel(leaves[m][1]) @ rho2 < 0 or vel(leaves[k][1]) @ rho2 < 0:
                        ok = False
                j += 1
            if not ok:
                break
        if not ok:
            break
        if draw(1, d, 0) < np.exp(min(0.0, sub_logw - logw)):
            prop = sub_prop
        logw = np.logaddexp(logw, sub_logw)
        sum_mom = sum_mom + sum(pp for _, pp in leaves)
        if fwd:
            pos = cur
        else:
            neg = cur
        if vel(neg[1]) @ sum_mom < 0 or vel(pos[1]) @ sum_mom < 0:
            break
    return prop, n_step


import pytest  # noqa: E402


@pytest.mark.parametrize("with_metric,extra", [(False, True), (True, True), (False, False)])
def test_batched_tree_equals_single_chain_restatement(emu_lib, with_metric, extra):  # noqa: F811
    _tree_vs_restatement(with_metric, extra, b"emu:host-TEST-ONLY")


@pytest.mark.gpu
@pytest.mark.parametrize("with_metric,extra", [(False, True), (True, True), (False, False)])
def test_batched_tree_equals_single_chain_restatement_hip(with_metric, extra):
    """The same on the MI355X: chmc_tree_step's device-side decisions (integrator errors, divergence, multinomial
    weights, proposal, no-U-turn termination incl. Mici's extra sub-tree checks) against the single-chain restatement,
    whose leapfrog steps run one chain at a time through the same library."""
    _tree_vs_restatement(with_metric, extra, b"hip:gfx950", T=12, S=16, R=5, B=5)  # 7-row blocks: the headline kernels


def _tree_vs_restatement(with_metric, extra, backend, T=6, S=4, R=2, B=3):
    from manifold_mcmc_for_diffusions_amd import _lib
    assert _lib.lib().chmc_backend() == backend
    from manifold_mcmc_for_diffusions_amd import example_models as em
    from manifold_mcmc_for_diffusions_amd.init import fhn_initial_states
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    from manifold_mcmc_for_diffusions_amd.dynamic import DynamicTransition, TreeUniforms, _ckpt_range
    assert [_ckpt_range(k)[1] for k in (0, 2, 4, 6)] == [0, 1, 1, 2]              # slot an even leaf is stored in
    assert [_ckpt_range(k) for k in (1, 3, 5, 7)] == [(0, 0), (0, 1), (1, 1), (0, 2)]  # slots an odd leaf is checked against
    y = em.simulate_fhn_observations(T, 0.2, 50, seed=5, sigma=0.1)
    eps, depth = 0.12, 4
    ctx = ChmcContext("fhn", 0.2, S, R, y[:, 0], sigma=0.1, num_chains=B)
    ctx1 = ChmcContext("fhn", 0.2, S, R, y[:, 0], sigma=0.1, num_chains=1)
    q, xo, _ = fhn_initial_states(em.fhn, 0.2, S, y, B, True, seed=7)
    ctx.set_state(q, None, xo, 0)
    W = None
    if with_metric:  # block metric on the global parameters: enters the integrator and the no-U-turn criterion
        a = np.random.default_rng(3).standard_normal((4, 4))
        M0 = a @ a.T / 4 + 0.5 * np.eye(4)
        ctx.set_metric(M0), ctx1.set_metric(M0)
        W = np.linalg.inv(M0)
    tr = DynamicTransition(ctx, eps, seed=11, max_tree_depth=depth, do_extra_subtree_checks=extra)
    n_total = 0
    for it in range(5):
        ctx.sample_momentum(11, it + 1)
        q0, p0, xo0, part = ctx.get_state()
        st = tr.sample(it)
        q1 = ctx.get_state()[0]
        un = TreeUniforms(11, it, B, 0, B)
        for c in range(B):
            prop, n = _reference_transition(ctx1, q0[c], p0[c], xo0[c], part,
                                            lambda kind, d, k, c=c: un.get(kind, d, k)[c], eps, depth, 1000.0, tr.solver, W,
                                            extra)
            assert n == st["n_step"][c], (it, c, n, st["n_step"][c])
            np.testing.assert_allclose(q1[c], prop, rtol=0, atol=1e-9)
            n_total += n
        ctx.switch_partition()
    assert n_total > 40  # trees of several doublings were built
    ctx.close(), ctx1.close()
