"""Golden vectors (tests/golden/*.npz, produced by the torch.func autodiff oracle) against
  * the C oracle (CPU),
  * the library's host logic through the TEST-ONLY emulation build (CPU),
  * the HIP library on the GPU (marked gpu).
Tolerances: operators 1e-9 relative (the golden side differentiates by reverse-mode autodiff, the others by analytic
adjoint / tangent sweeps: different rounding paths), leapfrog positions 1e-8 after two steps."""
import glob
import os
import numpy as np
import pytest
from oracle import c_oracle
from test_emu_logic import emu_lib  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
NAMES = [os.path.basename(f)[:-4] for f in FILES]
TOLS = dict(ctol=1e-9, ptol=1e-8, max_iters=50)


def load(name):
    g = dict(np.load(os.path.join(HERE, "golden", name + ".npz")))
    cfg = dict(model=str(g["model"]), T=int(g["T"]), S=int(g["S"]), R=None if int(g["R"]) < 0 else int(g["R"]),
               sigma="variable" if float(g["sigma"]) == -2.0 else None if float(g["sigma"]) < 0 else float(g["sigma"]),
               gaussian=bool(g["gaussian"]),
               obs_interval=float(g["obs_interval"]))
    return g, cfg


def close(a, b, tol, what):
    scale = max(1.0, float(np.max(np.abs(b))))
    err = float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / scale
    assert err < tol, f"{what}: rel err {err:.2e} >= {tol}"


@pytest.mark.parametrize("name", NAMES)
def test_c_oracle_matches_golden(name):
    g, cfg = load(name)
    osys = c_oracle.OracleSystem(cfg["model"], cfg["obs_interval"], cfg["S"], cfg["R"], g["y"], sigma=cfg["sigma"],
                                 use_gaussian_splitting=cfg["gaussian"])
    assert osys.num_partition == int(g["num_partition"]) and osys.rmax == int(g["rmax"])
    close(osys.generate_x_obs_seq(g["q"]), g["x_obs"], 1e-12, "x_obs_seq")
    for part in range(osys.num_partition):
        pre = f"p{part}_"
        c, du, dv = osys.jacob_constr_blocks(g["q_off"], g["x_obs"], part)
        cC, cD, ld, grad = osys.gram_ops(g["q_off"], g["x_obs"], part)
        Jw, JTl, Gil, nsc = osys.jacob_products(g["q_off"], g["x_obs"], part, g[pre + "w"], g[pre + "lam"])
        for k, v in (("c", c), ("dc_du", du), ("dc_dv", dv), ("chol_C", cC), ("chol_D", cD), ("grad", grad),
                     ("Jw", Jw), ("JTlam", JTl), ("Ginv_lam", Gil), ("nsc", nsc)):
            close(v, g[pre + k], 1e-9, f"{name} part {part} {k}")
        assert abs(ld - float(g[pre + "log_det"])) < 1e-9 * max(1.0, abs(ld))
    for sname, newton in (("newton", True), ("qn", False)):
        for dname, dirn in (("fwd", 1.0), ("bwd", -1.0)):
            ch = c_oracle.OracleChain(osys)
            ch.set(g["q"], g["p0"], g["x_obs"], 0)
            assert abs(ch.hamiltonian() - float(g["h0"])) < 1e-9 * max(1.0, abs(float(g["h0"])))
            pre = f"{sname}_{dname}_"
            st, itf, itb, _ = ch.step(dirn * 0.05, newton=newton, **TOLS)
            assert st == 0
            q1, p1, _, _ = ch.get()
            close(q1, g[pre + "q1"], 1e-9, pre + "q1")
            close(p1, g[pre + "p1"], 1e-8, pre + "p1")
            st, itf, itb, _ = ch.step(dirn * 0.05, newton=newton, **TOLS)
            assert st == 0 and (itf, itb) == tuple(int(v) for v in g[pre + "iters"])
            q2, p2, _, _ = ch.get()
            close(q2, g[pre + "q2"], 1e-8, pre + "q2")
            close(p2, g[pre + "p2"], 1e-7, pre + "p2")
            assert abs(ch.hamiltonian() - float(g[pre + "h2"])) < 1e-8 * max(1.0, abs(float(g[pre + "h2"])))


def check_library(name):
    from manifold_mcmc_for_diffusions_amd.context import ChmcContext
    g, cfg = load(name)
    ctx = ChmcContext(cfg["model"], cfg["obs_interval"], cfg["S"], cfg["R"], g["y"], sigma=cfg["sigma"],
                      use_gaussian_splitting=cfg["gaussian"], num_chains=2)
    two = lambda a: np.stack([a, a])  # noqa: E731  two identical chains: also checks chain independence
    rm = int(g["rmax"])
    for part in range(ctx.num_partition):
        pre = f"p{part}_"
        ctx.set_state(two(g["q_off"]), None, two(g["x_obs"]), part)
        du, dv = ctx.jacob_constr_blocks()
        cC, cD = ctx.chol_gram_blocks()
        for c in range(2):
            close(ctx.constr()[c], g[pre + "c"], 1e-9, "c")
            close(du[c], g[pre + "dc_du"], 1e-9, "dc_du")
            close(dv[c][:rm], g[pre + "dc_dv"], 1e-9, "dc_dv")
            close(cC[c], g[pre + "chol_C"], 1e-9, "chol_C")
            for b, blk in enumerate(ctx.blocks[part]):
                r = blk["nrows"]
                close(cD[c][b][:r, :r], g[pre + "chol_D"][b][:r, :r], 1e-9, "chol_D")
            assert abs(ctx.log_det_sqrt_gram()[c] - float(g[pre + "log_det"])) < 1e-9 * max(1.0, abs(float(g[pre + "log_det"])))
            close(ctx.grad_log_det_sqrt_gram()[c], g[pre + "grad"], 1e-9, "grad")
            close(ctx.lmult_by_jacob_constr(two(g[pre + "w"]))[c], g[pre + "Jw"], 1e-9, "Jw")
            close(ctx.rmult_by_jacob_constr(two(g[pre + "lam"]))[c], g[pre + "JTlam"], 1e-9, "JTlam")
            close(ctx.lmult_by_inv_gram(two(g[pre + "lam"]))[c], g[pre + "Ginv_lam"], 1e-9, "Ginv_lam")
            close(ctx.normal_space_component(two(g[pre + "w"]))[c], g[pre + "nsc"], 1e-9, "nsc")
    for sname, newton in (("newton", True), ("qn", False)):
        ctx.set_state(two(g["q"]), two(g["p0"]), two(g["x_obs"]), 0)
        assert np.abs(ctx.hamiltonian()[:, 0] - float(g["h0"])).max() < 1e-9 * max(1.0, abs(float(g["h0"])))
        dt = np.array([0.05, -0.05])  # chain 0 forward, chain 1 backward in the same call
        r1 = ctx.leapfrog_step(dt, newton=newton, constraint_tol=1e-9, position_tol=1e-8, max_iters=50)
        q1, p1, _, _ = ctx.get_state()
        r2 = ctx.leapfrog_step(dt, newton=newton, constraint_tol=1e-9, position_tol=1e-8, max_iters=50)
        q2, p2, _, _ = ctx.get_state()
        assert (r1["status"] == 0).all() and (r2["status"] == 0).all()
        for c, dname in enumerate(("fwd", "bwd")):
            pre = f"{sname}_{dname}_"
            close(q1[c], g[pre + "q1"], 1e-9, pre + "q1")
            close(p1[c], g[pre + "p1"], 1e-8, pre + "p1")
            close(q2[c], g[pre + "q2"], 1e-8, pre + "q2")
            close(p2[c], g[pre + "p2"], 1e-7, pre + "p2")
            assert (int(r2["iters_fwd"][c]), int(r2["iters_bwd"][c])) == tuple(int(v) for v in g[pre + "iters"])
    ctx.close()


@pytest.mark.parametrize("name", NAMES)
def test_host_logic_matches_golden(emu_lib, name):  # noqa: F811
    check_library(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_golden(name):
    from manifold_mcmc_for_diffusions_amd import _lib
    assert _lib.lib().chmc_backend() == b"hip:gfx950"
    check_library(name)
