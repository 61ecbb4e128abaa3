#!/usr/bin/env python3
"""The reference's notebook experiment (FitzHugh-Nagumo_example.ipynb) on an MI355X: FitzHugh-Nagumo model with the
notebook's priors, 100 noiseless observations of the first component every 0.5 time units, 25 steps per observation,
5 observations per sub-sequence, Gaussian splitting, Newton retraction -- the data are regenerated from the notebook's
seed (legacy RandomState(20200710), 5006 standard normals).  The posterior summary is compared with the table the
notebook prints (tests/golden/reference_data/notebook_posterior_table.json), which is the only known-answer material
the reference holds for this path.  The notebook samples with Mici's dynamic multinomial transition; here the batched
static-trajectory sampler is used (a different Markov kernel for the same posterior), with many more chains.

With `dynamic` as sixth argument the batched dynamic (no-U-turn, multinomial) transition of dynamic.py is used, the
counterpart of the notebook's own transition.

usage: fhn_notebook_posterior.py [chains] [iterations] [warm-up] [steps per trajectory] [output dir] [static|dynamic]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd import example_models as em  # noqa: E402
from manifold_mcmc_for_diffusions_amd.context import ChmcContext  # noqa: E402
from manifold_mcmc_for_diffusions_amd.sampling import sample_static_chmc  # noqa: E402
from manifold_mcmc_for_diffusions_amd.traces import summarize  # noqa: E402

TABLE = os.path.join(ROOT, "tests", "golden", "reference_data", "notebook_posterior_table.json")


def notebook_data():
    """Cells 23-27: q_ref from the legacy generator, y_seq_ref by simulating the model forward."""
    m = em.fhn_nb
    T, S, dt_obs = 100, 25, 0.5
    q_ref = np.random.RandomState(20200710).standard_normal(size=4 + 2 + T * S * 2)
    z = m.generate_z(q_ref[:4])
    x_0 = m.generate_x_0(z, q_ref[4:6])
    x_seq = em.generate_x_seq(m, z, x_0, q_ref[6:].reshape((-1, 2)), dt_obs / S)
    return dict(T=T, S=S, obs_interval=dt_obs, y=x_seq[S - 1::S, 0].copy(), z_ref=z, x_0_ref=x_0, q_ref=q_ref)


def run(num_chains=64, n_iter=700, n_warm=200, n_step=24, out_dir=None, seed=20200710, verbose=True,
        transition="static"):
    d = notebook_data()
    m = em.fhn_nb
    rng = np.random.default_rng(seed)

    def draws(n):  # cells 37-39: the random inputs of find_initial_state_by_linear_interpolation
        return (rng.standard_normal((n, 4)), rng.standard_normal((n, 2)),
                np.concatenate([np.broadcast_to(d["y"][None, :, None], (n, d["T"], 1)),
                                0.5 * rng.standard_normal((n, d["T"], 1))], -1))

    def context(n):
        return ChmcContext("fhn_nb", d["obs_interval"], d["S"], 5, d["y"], sigma=None, use_gaussian_splitting=True,
                           num_chains=n)

    # Prior draws with a small time-scale separation eps make the explicit step unstable at this step size (the
    # forward map amplifies round-off by > 5 per step): the interpolated path then does not reproduce itself and the
    # state is useless as a start, for the reference as much as here (it has no check; its two chains happened to be
    # fine).  Candidates are screened on the device and the first `num_chains` usable ones are kept.
    cand = context(3 * num_chains)
    u, v_0, x_obs_init = draws(3 * num_chains)
    cand.init_by_linear_interpolation(u, v_0, x_obs_init)
    usable = np.isfinite(cand.hamiltonian()[:, 0]) & (np.abs(cand.constr()).max(1) < 1e-8)
    # ... and a short pilot run weeds out the starts from which no trajectory is ever accepted (Gram matrix so
    # ill-conditioned that the energy is noise): a chain that has not moved in 20 short transitions never will
    pilot = sample_static_chmc(cand, 20, 4, 0.05, seed=seed + 1, n_adapt=20, jitter_length=True)
    usable &= (np.diff(pilot["heads"][:, :, 0], axis=0) != 0).sum(0) >= 5
    cand.close()
    keep = np.flatnonzero(usable)[:num_chains]
    if verbose:
        print(f"initial states: {usable.sum()} of {usable.size} prior draws usable, keeping {keep.size}")
    assert keep.size == num_chains, "not enough usable initial states"
    ctx = context(num_chains)
    ctx.init_by_linear_interpolation(u[keep], v_0[keep], x_obs_init[keep])

    def trace_func(head, ham):  # cell 41
        z = m.generate_z(head[:, :4])
        return {"σ": z[:, 0], "ϵ": z[:, 1], "γ": z[:, 2], "β": z[:, 3], "x_0": m.generate_x_0(z, head[:, 4:6])}

    tmp = out_dir or os.path.join(ROOT, "gpurun_out", "fhn_notebook_run")
    t0 = time.time()
    if transition == "dynamic":
        from manifold_mcmc_for_diffusions_amd.dynamic import sample_dynamic_chmc
        res = sample_dynamic_chmc(ctx, n_iter, 0.1, seed=seed, n_adapt=n_warm, trace_dir=tmp, trace_func=trace_func,
                                  solver=dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10,
                                              max_iters=50, reverse_check_tol=2e-8),
                                  callback=(lambda it, h, a, e, st: (it % 50 == 0) and print(
                                      f"  iter {it:4d} accept {a:.2f} step {e:.3f} n_step {st['n_step'].mean():.1f} "
                                      f"integrator errors {st['integrator_error'].mean():.2f}", flush=True)) if verbose else None)
        res["fail_rate"] = res["integrator_error"]
    else:
      res = sample_static_chmc(ctx, n_iter, n_step, 0.1, seed=seed, n_adapt=n_warm, trace_dir=tmp, trace_func=trace_func,
                             jitter_length=True,
                             solver=dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10,
                                         max_iters=50, reverse_check_tol=2e-8),  # cell 33
                             callback=(lambda it, h, a, e: (it % 50 == 0) and print(
                                 f"  iter {it:4d} accept {a:.2f} step {e:.3f}", flush=True)) if verbose else None)
    el = time.time() - t0
    ctx.close()
    ref = json.load(open(TABLE))
    # A chain that (almost) never moves in the main phase is not exploring a mode, it is stuck on an unusable start that
    # the pilot missed (healthy chains move in 80-100 % of their transitions, stuck ones in 0-3 %); chains that moved in
    # fewer than 10 % of the main transitions are left out of the summary and counted.
    tr = {k: np.load(f) for k, f in res["trace_files"].items()}
    moved = np.diff(tr["σ"][:, n_warm:], axis=1) != 0
    moving = moved.mean(1) >= 0.1
    # ... and a chain that is released from such a start only during the main phase and then travels to the bulk is still
    # in its transient.  Burn-in check per chain (Geweke-style): the mean of a parameter over the first fifth of the main
    # phase against its mean over the second half, in units of the pooled within-chain standard deviation of the moving
    # chains' second halves; more than 4 of those for any of the four parameters -> left out as well (healthy chains: < 1).
    n_main = tr["σ"].shape[1] - n_warm
    drift = np.zeros(num_chains)
    for k in ("σ", "ϵ", "γ", "β"):
        v = tr[k][:, n_warm:]
        late = v[:, n_main // 2:]
        sd = np.sqrt(np.mean(late[moving].var(1))) if moving.any() else 1.0
        drift = np.maximum(drift, np.abs(v[:, :max(n_main // 5, 1)].mean(1) - late.mean(1)) / max(sd, 1e-300))
    moving &= drift <= 4.0
    sm = summarize({k: v[moving][:, n_warm:] for k, v in tr.items()})
    if verbose and not moving.all():
        # diagnosis of the chains left out: where their parameters sit and how their transitions ended
        sm_all = summarize({k: v[:, n_warm:] for k, v in tr.items()})
        print("chains left out of the summary (moved in < 10 % of the main transitions, or still drifting: first fifth vs second half > 4 sd):")
        oc, od = res.get("chain_outcomes"), res.get("chain_outcomes_dynamic")
        for c in np.flatnonzero(~moving):
            line = f"  chain {c:3d}: sigma {tr['σ'][c, -1]:.3f} eps {tr['ϵ'][c, -1]:.4f} gamma {tr['γ'][c, -1]:.3f} beta {tr['β'][c, -1]:.3f}"
            if oc is not None:
                line += ("  transitions: accepted %d, rejected %d, not converged %d, diverged %d, non-reversible %d"
                         % tuple(oc[c]))
            if od is not None:
                line += ("  trees: moved %d, not moved %d, ended by integrator error %d, by divergence %d, no leaf at all %d, "
                         "non-finite root energy %d; mean leaves %.1f" % (*od[c], res["per_chain"]["n_step"][c]))
            print(line)
        print("  posterior means over ALL chains (no selection): " + ", ".join(
            f"{k} {sm_all['mean'][k]:.3f}" for k in ("σ", "ϵ", "γ", "β", "x_0[0]", "x_0[1]")))
    rows = []
    for k in ("σ", "ϵ", "γ", "β", "x_0[0]", "x_0[1]"):
        mc = sm["sd"][k] / np.sqrt(max(sm["ess_bulk"][k], 1.0))
        zscore = (sm["mean"][k] - ref["mean"][k]) / np.hypot(mc, ref["mcse_mean"][k])
        rows.append(dict(var=k, mean=sm["mean"][k], sd=sm["sd"][k], r_hat=sm["r_hat"][k], ess=sm["ess_bulk"][k],
                         ref_mean=ref["mean"][k], ref_sd=ref["sd"][k], z=zscore))
    if verbose and transition == "dynamic":
        pc = res["per_chain"]
        print("  sampler statistics over the chains of the summary only (the pooled figures below average the left-out chains in): "
              f"accept {pc['accept'][moving].mean():.2f}, leaves per tree {pc['n_step'][moving].mean():.1f}, trees ended by an "
              f"integrator error or divergence {pc['err'][moving].mean():.3f}")
    if verbose:
        print(f"{int(moving.sum())} of {num_chains} chains moving in the main phase (summary over those)")
        print(f"{num_chains} chains x {n_iter} transitions x <= {n_step} steps in {el:.1f} s; final step size "
              f"{res['final_step_size']:.3f}, accept {res['accept_stat'][n_warm:].mean():.2f}, failed trajectories "
              f"{res['fail_rate'][n_warm:].mean():.3f}" + (f", mean tree size {res['n_step'][n_warm:].mean():.1f} steps"
                                                             if transition == "dynamic" else ""))
        print("  var      mean (notebook)     sd (notebook)     r_hat   ess    z = diff / combined mcse")
        for r in rows:
            print(f"  {r['var']:7s} {r['mean']:7.3f} ({r['ref_mean']:6.3f})   {r['sd']:6.3f} ({r['ref_sd']:5.3f})   "
                  f"{r['r_hat']:.3f} {r['ess']:6.0f}   {r['z']:+.2f}")
        print("  data-generating values: sigma %.3f eps %.3f gamma %.3f beta %.3f x_0 %s" % (
            *d["z_ref"], np.round(d["x_0_ref"], 3)))
    return rows, res, int(moving.sum())


if __name__ == "__main__":
    a = sys.argv[1:]
    run(*(int(x) for x in a[:4]), out_dir=a[4] if len(a) > 4 and a[4] != "-" else None,
        transition=a[5] if len(a) > 5 else "static")
