#!/usr/bin/env python3
"""Generates the golden vectors in this directory with the Python (torch.func fp64 autodiff) oracle, i.e. the
operator-for-operator restatement of sde/mici_extensions.py in oracle/py/system.py.

The reference itself cannot run in the build container (jax / mici / symnum are not installed) and
ships no tests or golden vectors, so these fixtures pin the C oracle and the HIP library to the autodiff
restatement, not to outputs of the reference ("parity unpinned").   Run:  python tests/golden/generate.py
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.py import models as omodels, system as osys  # noqa: E402
from helpers import random_q  # noqa: E402

CASES = [
    # name, model, T, S, R, noisy, gaussian, obs_interval
    ("fhn_noisy_std", "fhn", 6, 4, 2, True, False, 0.2),
    ("fhn_noisy_gauss", "fhn", 6, 4, 2, True, True, 0.2),
    ("fhn_noiseless_std", "fhn", 7, 5, 3, False, False, 0.2),
    ("fhn_noisy_r5", "fhn", 12, 6, 5, True, False, 0.2),
    ("sir_single_block", "sir", 5, 6, None, True, False, 0.25),
    ("sir_partitioned", "sir", 6, 8, 2, True, False, 0.25),
    # the notebook's model (FitzHugh-Nagumo_example.ipynb): noiseless observations every 0.5, Gaussian splitting, R = 5
    ("fhn_nb_noiseless_gauss", "fhn_nb", 7, 25, 5, False, True, 0.5),
    ("fhn_nb_noisy_std", "fhn_nb", 6, 4, 2, True, False, 0.2),
    # variable observation noise: sigma = generate_σ_y(u) = exp(u[4]), dim_u = 5 (scripts/sir_model_chmc_experiment.py:44-77)
    ("sir_varsigma_single_block", "sir", 5, 6, None, "variable", False, 0.25),
    ("sir_varsigma_partitioned", "sir", 6, 8, 2, "variable", False, 0.25),
    ("fhn_varsigma_gauss", "fhn", 6, 4, 2, "variable", True, 0.2),
]
TOLS = dict(constraint_tol=1e-9, position_tol=1e-8, max_iters=50)


def rowslot(jac, rmax, nv):
    du = torch.cat([b.reshape(-1, b.shape[-1]) for b in jac[0]]).numpy()
    dv = np.zeros((rmax, nv))
    col = 0
    for b in jac[1]:
        b = b.numpy()
        if b.ndim == 2:
            b = b[None]
        for m in range(b.shape[0]):
            r, nc = b[m].shape
            dv[:r, col:col + nc] = b[m]
            col += nc
    return du, dv


def pad_chol(chol, rmax):
    out = []
    for ch in chol[1]:
        ch = ch.numpy()
        if ch.ndim == 2:
            ch = ch[None]
        for m in range(ch.shape[0]):
            p = np.zeros((rmax, rmax))
            r = ch[m].shape[0]
            p[:r, :r] = ch[m]
            out.append(p)
    return np.stack(out)


def main():
    only = set(sys.argv[1:])  # regenerate only the named cases (the committed files of the others stay as they are)
    for name, mname, T, S, R, noisy, gaussian, oi in CASES:
        if only and name not in only:
            continue
        rng = np.random.default_rng(11 if mname == "sir" else sum(map(ord, name)))
        model = omodels.MODELS[mname]
        var_sigma = noisy == "variable"
        noisy = bool(noisy)
        sigma = "variable" if var_sigma else ((0.1 if mname in ("fhn", "fhn_nb") else 1.0) if noisy else None)
        q = random_q(mname, T, S, noisy, 1, rng, var_sigma=var_sigma)[0]
        sys0 = osys.make_system(model, oi, S, R, np.zeros((T, 1)), sigma=sigma, use_gaussian_splitting=gaussian)
        xo = sys0._generate_x_obs_seq(osys.T(q)).numpy()
        sig0 = float(np.exp(q[model.dim_z])) if var_sigma else sigma
        y = model.obs_func(osys.T(xo)).numpy() + (sig0 * q[-T:, None] if noisy else 0.0)
        sysm = osys.make_system(model, oi, S, R, y, sigma=sigma, use_gaussian_splitting=gaussian)
        out = dict(model=mname, T=T, S=S, R=-1 if R is None else R, noisy=noisy, gaussian=gaussian, obs_interval=oi,
                   sigma=-1.0 if sigma is None else (-2.0 if var_sigma else sigma), y=y[:, 0], q=q, x_obs=xo)  # -2: variable
        rmax = max(int(b.shape[-2]) for p in range(sysm.num_partition)
                   for b in sysm._jacob_constr_blocks(osys.T(q), osys.T(xo), p)[1])
        nv = model.dim_v_0 + T * S * model.dim_v
        out["rmax"], out["num_partition"] = rmax, sysm.num_partition
        # off-manifold point for the op-level vectors (non-zero constraint values)
        q_off = q + 0.01 * rng.standard_normal(q.shape)
        out["q_off"] = q_off
        for part in range(sysm.num_partition):
            st = osys.ConditionedDiffusionHamiltonianState(q_off, xo, part)
            jac = sysm.jacob_constr_blocks(st)
            chol = sysm.chol_gram_blocks(st)
            du, dv = rowslot(jac, rmax, nv)
            w = rng.standard_normal(q.shape)
            lam = rng.standard_normal(du.shape[0])
            pre = f"p{part}_"
            out.update({
                pre + "c": sysm.constr(st), pre + "dc_du": du, pre + "dc_dv": dv, pre + "chol_C": chol[0].numpy(),
                pre + "chol_D": pad_chol(chol, rmax), pre + "log_det": sysm.log_det_sqrt_gram(st),
                pre + "grad": sysm.grad_log_det_sqrt_gram(st), pre + "w": w, pre + "lam": lam,
                pre + "Jw": sysm._lmult_by_jacob_constr(*jac, osys.T(w)).numpy(),
                pre + "JTlam": sysm._rmult_by_jacob_constr(*jac, osys.T(lam)).numpy(),
                pre + "Ginv_lam": sysm._lmult_by_inv_gram(*jac, *chol, osys.T(lam)).numpy(),
                pre + "nsc": sysm.normal_space_component(st, w),
            })
        # leapfrog steps from the on-manifold point, both solvers, +dt and -dt
        st0 = osys.ConditionedDiffusionHamiltonianState(q, xo, 0)
        st0.mom = sysm.sample_momentum(st0, rng)
        out["p0"] = st0.mom
        out["h0"] = sysm.h(st0)
        for sname, solver in (("newton", osys.jitted_solve_projection_onto_manifold_newton),
                              ("qn", osys.jitted_solve_projection_onto_manifold_quasi_newton)):
            for dname, dirn in (("fwd", 1), ("bwd", -1)):
                integ = osys.ConstrainedLeapfrogIntegrator(sysm, step_size=0.05, projection_solver=solver,
                                                           projection_solver_kwargs=TOLS)
                s = st0.copy()
                s.dir = dirn
                s1 = integ.step(s)
                s2 = integ.step(s1)
                pre = f"{sname}_{dname}_"
                out.update({pre + "q1": s1.pos, pre + "p1": s1.mom, pre + "q2": s2.pos, pre + "p2": s2.mom,
                            pre + "h2": sysm.h(s2), pre + "iters": np.array(integ.last_iters)})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "Q", q.size, "rmax", rmax)


if __name__ == "__main__":
    main()
