#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: constrained-leapfrog steps/sec (all chains), FHN noisy-obs, 400 sub-steps.

One "step" = one batched ConstrainedLeapfrogIntegrator.step over the chains resident on a GPU (256 per GPU,
BASELINE.json configs[1]); weak scaling over GPUs (chains shard, no data-path collective, one RCCL gather of the
traced samples).  Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement".

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: spawns the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...

Other BASELINE.json configurations (parity-test cases; their lines are kept under profiles/):
  --config fhn_noiseless | sir      --solver quasi-newton      --splitting gaussian
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "constrained-leapfrog steps/sec (all chains), FHN noisy-obs 400 sub-steps"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # fp64 vector peak: 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 flop x 2.4 GHz

# Algorithmic flops of the SDE recursion per time step (+-30 % estimates): forward step,
# Jacobian sweep = shared part + per live adjoint row.  (SIR: scaled from its X = V = 3 one-step map.)
FLOPS = {"fhn": dict(fwd=32.0, jac_shared=15.0, jac_row=28.0), "sir": dict(fwd=60.0, jac_shared=40.0, jac_row=52.0)}
# what binds each kernel class on gfx950: "hbm" classes stream stored rows / state vectors,
# "fp64_valu" classes regenerate the Jacobian in registers and are bound by fp64 vector issue
BOUND = {"newton_blk": "fp64_valu", "state_blk": "fp64_valu", "grad_log_det_blk": "fp64_valu", "constr": "fp64_valu",
         "update": "hbm", "jacob_vec": "hbm", "elementwise": "hbm", "solve_chain": "latency", "sym_blk": "latency",
         "other": "latency"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="fhn_noisy", choices=["fhn_noisy", "fhn_noiseless", "sir"],
                    help="fhn_noisy = BASELINE.json configs[1] (the metric); fhn_noiseless = configs[2]; sir = configs[3]")
    ap.add_argument("--solver", default="newton", choices=["newton", "quasi-newton"])
    ap.add_argument("--splitting", default="standard", choices=["standard", "gaussian"])
    ap.add_argument("--chains-per-gpu", type=int, default=256)
    ap.add_argument("--num-steps-per-obs", type=int, default=None, help="default 400 (FHN) / 200 (SIR)")
    ap.add_argument("--num-obs", type=int, default=100, help="FHN only (rehearsals)")
    ap.add_argument("--step-size", type=float, default=None)
    ap.add_argument("--traj-len", type=int, default=16, help="leapfrog steps between momentum refreshes")
    ap.add_argument("--burn-iters", type=int, default=5)
    ap.add_argument("--burn-steps", type=int, default=16)
    ap.add_argument("--burn-step-size", type=float, default=None)
    ap.add_argument("--lockstep", action="store_true",
                    help="one chmc_leapfrog_step call per step with the trajectory bookkeeping on the host instead of one "
                         "chmc_leapfrog_steps call per trajectory (same batched steps underneath)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="record no HIP events in the timed region")
    ap.add_argument("--profile-stride", type=int, default=1,
                    help="timed region: HIP events around every n-th launch of the dominant kernel (an event pair "
                         "costs the stream ~30 us, timing every launch slows the region by ~5 %%)")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--cpu-autodiff-baseline", action="store_true",
                    help="also time ONE leapfrog step of the torch.func autodiff restatement (oracle/py, the operator-for-"
                         "operator analogue of the reference's JAX path, eager and unjitted) at BASELINE.json configs[0]'s "
                         "shape (S = 50) on one core: about two minutes")
    ap.add_argument("--data-steps-per-obs", type=int, default=10000, help="fine grid of the simulated FHN data")
    ap.add_argument("--collective", default="torch", choices=["torch", "cabi"],
                    help="route of the one gather of samples: torch.distributed (backend nccl = RCCL) or the library's own "
                         "C-ABI collective (chmc_comm_unique_id on rank 0, the 128-byte id broadcast through torch.distributed, "
                         "chmc_comm_init, chmc_gather_samples = ncclAllGather between device buffers)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (configs[1], one GPU) only: skip the short legs of the other single-GPU BASELINE shapes "
                         "that follow the headline's timed region (config.other_configs)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region is repeated this many times AFTER the official one (same K steps each) for the "
                         "run-to-run spread (config.value_repeats); `value` is always the first, official region")
    return ap.parse_args()


def config_key(config, S, B):
    """Key of a workload in profiles/traffic.json: the bench configuration plus the shape when it is not the default one."""
    k = config
    if S != (200 if config == "sir" else 400):
        k += f"_s{S}"
    if B != 256:
        k += f"_b{B}"
    return k


OTHER_CONFIGS = [  # the other single-GPU BASELINE.json shapes, run as short child legs after the headline (name, bench arguments)
    ("configs[2] fhn_noiseless S=400 x 256", ["--config", "fhn_noiseless"]),
    ("configs[3] shard: sir S=200 x 256", ["--config", "sir"]),
    ("configs[4] shard: fhn_noisy S=800 x 512", ["--config", "fhn_noisy", "--num-steps-per-obs", "800", "--chains-per-gpu", "512"]),
]


def run_other_configs(a):
    """Short legs of the other single-GPU BASELINE shapes, each in a fresh child process (its own context; this process
    keeps its own): 16 warm-up + 32 timed steps (whole trajectories of 16: the single-block layouts run a trajectory as one
    launch per chain), no CPU baseline.  Returns {name: summary or {"error": ...}}."""
    import subprocess
    out = {}
    for name, extra in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-configs", "--steps", "32",
               "--warmup", "16", "--repeats", "0"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": f"exit code {r.returncode}: {r.stderr.strip()[-300:]}"}
                continue
            d = json.loads(line[-1])
            c, rf = d["config"], d["roofline"]
            out[name] = {
                "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "workload": c["workload"], "newton_rounds_per_step": c["newton_rounds_per_step"],
                "mean_newton_iters_fwd_plus_bwd": c["mean_newton_iters_fwd_plus_bwd"],
                "step_success_rate": c["step_success_rate"], "launches_per_step": c["launches_per_step"],
                "nominal_steps_per_s": c.get("nominal_steps_per_s"),
                "roofline": {k: rf.get(k) for k in ("kernel", "bound", "frac", "achieved", "peak", "unit", "traffic",
                                                    "avg_launch_ms", "traffic_source")},
                "whole_step_hbm_frac": c.get("whole_step_hbm_frac"), "wall_s": round(time.perf_counter() - t0, 1),
                "command": "python bench.py " + " ".join(cmd[2:]),
            }
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)[:300]}
    return out


def spawn_ranks(a):
    """`--gpus N` outside a torchrun environment: start the N ranks as child processes (one per GPU, RCCL) BEFORE this
    process touches torch.cuda or the HIP library, relay their output and exit with their code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def cpu_baseline(wl, model, q, p, xo, part, dt, n_steps, solver, gaussian):
    """The C oracle (kind "port": the reference itself cannot run here) on the host cores, one chain per thread
    (ctypes releases the GIL), on a bounded sample of the same workload: `cores` chains x n_steps leapfrog steps from
    the chains' post-burn-in states.  `value` is the -O3 -march=native build compiled on this machine; the portable -O2
    build the parity tests use is timed beside it on a quarter of the sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(len(q), avail, 16))  # a one-GPU box has a 16-core CPU share
    kw = dict(newton=solver["newton"], ctol=solver["constraint_tol"], ptol=solver["position_tol"],
              dtol=solver["divergence_tol"], max_iters=solver["max_iters"], rev_tol=solver["reverse_check_tol"])

    def timed(kind, n):
        c_oracle.select_build(kind)
        osys = c_oracle.OracleSystem(model, wl.obs_interval, wl.S, wl.R, wl.y[:, 0], sigma=wl.sigma,
                                     use_gaussian_splitting=gaussian)
        chains = []
        for c in range(cores):
            ch = c_oracle.OracleChain(osys)
            ch.set(q[c], p[c], xo[c], part)  # state caches evaluated here, outside the timed region
            chains.append(ch)

        def run(ch):
            for _ in range(n):
                ch.step(dt, **kw)
            return n

        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(run, chains))
        el = time.perf_counter() - t0
        # one chain alone on one core (the reference takes its op timings pinned to a single core,
        # run_fhn_model_noiseless_obs_experiments.sh:115): a quarter of the steps, continuing from chain 0's state
        n1 = max(n // 4, 4)
        aff = None
        try:  # pinned to one core, as the reference's `taskset -c 0` does for its op timings
            aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {min(aff)})
        except (AttributeError, OSError):
            aff = None
        t1 = time.perf_counter()
        for _ in range(n1):
            chains[0].step(dt, **kw)
        single = n1 / (time.perf_counter() - t1)
        if aff is not None:
            os.sched_setaffinity(0, aff)
        return done / el, single, el

    try:
        v_nat, single_nat, el = timed("native", n_steps)
        flags = c_oracle.CFLAGS
        v_port, single_port, _ = timed("portable", max(n_steps // 4, 4))
    finally:
        c_oracle.select_build("portable")
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model_name = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": v_nat, "unit": "steps/s", "cores": cores, "kind": "port", "cpu_model": model_name,
            "per_core": v_nat / cores, "single_core_alone": single_nat, "single_core_pinned": True,
            "compiler_flags": "gcc " + flags,
            "portable_O2_build": {"value": v_port, "single_core_alone": single_port,
                                  "compiler_flags": "gcc " + c_oracle.PORTABLE_CFLAGS},
            "sample": f"{cores} chains x {n_steps} leapfrog steps from post-burn-in states, C oracle "
                      f"(oracle/c/chmc_oracle.c, gcc {flags}), one chain per host thread, {el:.1f} s"}


def cpu_autodiff_baseline(T=100):
    """SURVEY.md 8(d) baseline (ii): the Python restatement with torch.func fp64 autodiff standing in for JAX (jacrev of
    a scan, grad through jacrev), one chain, one core, configs[0]'s shape (FHN noisy, T = 100, S = 50; the default line runs a
    bounded sample of it, T = 20 observations = 4 sub-sequences, about 20 s).  Eager PyTorch interprets every time step in
    Python, which jitted XLA does not: a lower bound on what the reference's CPU path does, quoted for completeness and not
    comparable with the compiled figures."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.py import models as omodels, system as osys
    from helpers import make_case
    torch.set_num_threads(1)
    case = make_case("fhn", T, 50, 5, True, B=1, seed=1)
    sysm = osys.make_system(omodels.fhn, 0.2, 50, 5, case["y"][:, None], sigma=0.1)
    st = osys.ConditionedDiffusionHamiltonianState(case["q"][0], case["x_obs"][0], 0)
    st.mom = sysm.sample_momentum(st, np.random.default_rng(0))
    integ = osys.ConstrainedLeapfrogIntegrator(
        sysm, step_size=0.05, projection_solver=osys.jitted_solve_projection_onto_manifold_newton,
        projection_solver_kwargs=dict(constraint_tol=1e-9, position_tol=1e-8, max_iters=50))
    t0 = time.perf_counter()
    integ.step(st)
    el = time.perf_counter() - t0
    return {"value": 1.0 / el, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": f"1 chain x 1 leapfrog step, torch.func autodiff restatement (oracle/py), FHN noisy T={T} S=50 "
                      f"({'configs[0] shape' if T == 100 else 'configs[0] with ' + str(T) + ' of its 100 observations'}, dim_q "
                      f"{case['q'].shape[1]}), eager PyTorch on one core, {el:.0f} s"}


def _finite(o):
    """NaN / inf are not JSON: unmeasured figures are printed as null."""
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, float) and not np.isfinite(o):
        return None
    return o


def lib_sha256():
    from manifold_mcmc_for_diffusions_amd import _lib
    h = hashlib.sha256()
    with open(_lib._SO, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def class_model(ctx, wl, model, newton):
    """Per kernel class: algorithmic bytes and flops per chain of ONE launch (SURVEY.md 8d operator decomposition)."""
    nnz, Q = wl.nnz(), ctx.Q
    f = FLOPS[model]
    blocks = ctx.blocks[ctx.partition]
    n_s = ctx.T * ctx.S
    live = 0.0  # sum over time steps of the live adjoint rows (observation row i is zero after its own interval)
    gram_ns = gram_sym = 0.0
    for b in blocks:
        x_rows = b["nrows"] - b["ny"]
        for m in range(b["nobs"]):
            live += ctx.S * (max(b["ny"] - m, 0) + x_rows)
        gram_ns += 2.0 * b["nrows"] ** 2 * b["ncols"]
        gram_sym += 1.0 * b["nrows"] * (b["nrows"] + 1) * b["ncols"]
    jac = f["jac_shared"] * n_s + f["jac_row"] * live
    # structurally non-zero entries of the stored dc/dv rows (the reference's dense blocks hold `nnz`; the row-slot layout
    # lets the kernels skip an observation row after its own interval) + the observation-noise columns
    nnz_live = live * ctx.V + (ctx.T if ctx.noisy else 0)
    # the Newton-loop and projection passes read the COMPACT rows: X V doubles per step (Slots::PB) + the frames
    n_pb = n_s * ctx.X * ctx.V + (ctx.T if ctx.noisy else 0)
    n_rows = min(nnz_live, n_pb)
    hbm = {"update": 8.0 * (n_rows + 2 * Q), "jacob_vec": 8.0 * (n_rows + 2 * Q), "elementwise": 8.0 * 3 * Q,
           # traj + v + compact rows read per step; nothing per-step is written
           "newton_blk": 8.0 * (n_s * ctx.X + Q + n_rows)}
    cm = _class_model_operator(nnz, Q, jac, gram_ns, gram_sym, f, n_s)
    for k, v in hbm.items():
        cm[k]["hbm_bytes"] = v  # what the fused kernel has to move across HBM (<= the operator-level figure)
    return cm


def _class_model_operator(nnz, Q, jac, gram_ns, gram_sym, f, n_s):
    return {
        # newton_blk = constr (Q) + jacob_constr_blocks (Q + nnz written) + lu_jacob_product_blocks (2 nnz)
        "newton_blk": dict(bytes=8.0 * (3 * nnz + 2 * Q), flops=jac + gram_ns),
        # state_blk  = constr (Q) + jacob (Q + nnz) + chol_gram_blocks (nnz)
        "state_blk": dict(bytes=8.0 * (2 * nnz + 2 * Q), flops=jac + gram_sym),
        "grad_log_det_blk": dict(bytes=8.0 * (4 * nnz + 2 * Q), flops=2.0 * (jac + gram_ns)),  # two sweeps per evaluation
        "update": dict(bytes=8.0 * (nnz + 6 * Q), flops=2.0 * nnz),
        "jacob_vec": dict(bytes=8.0 * (nnz + Q), flops=2.0 * nnz),
        "constr": dict(bytes=8.0 * 2 * Q, flops=f["fwd"] * n_s),
        "elementwise": dict(bytes=8.0 * 3 * Q, flops=2.0 * Q),
    }


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    from manifold_mcmc_for_diffusions_amd import distributed as D
    emu = os.environ.get("CHMC_BENCH_EMU_LIB")  # TEST-ONLY: CPU rehearsal of the launcher / multi-rank path (tests/)
    rank, local_rank, world = D.init_process_group("gloo" if emu else None)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    import torch
    from manifold_mcmc_for_diffusions_amd import _lib
    if emu:
        _lib._LIB = _lib._bind(C.CDLL(emu))
        dev, devno = torch.device("cpu"), 0
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        devno = int(os.environ.get("CHMC_BENCH_DEVICE", local_rank))  # override only for single-GPU rehearsals
        torch.cuda.set_device(devno)
        dev = torch.device("cuda", devno)
    sync = (lambda: None) if emu else torch.cuda.synchronize
    from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload, SirWorkload
    L = _lib.lib()
    B = a.chains_per_gpu
    gaussian = a.splitting == "gaussian"
    model = "sir" if a.config == "sir" else "fhn"
    if a.config == "sir":
        S = a.num_steps_per_obs or 200
        wl = SirWorkload(B, num_steps_per_obs=S, device=devno, chain_offset=rank * B, total_chains=world * B,
                         use_gaussian_splitting=gaussian)
        step_size = a.step_size if a.step_size is not None else 0.25
        workload = (f"SIR boarding-school data, T=14 obs, S={S} steps/obs, one sub-sequence of R=14 (dense 14x14 Gram), "
                    f"sigma_y=1, {B} chains per GPU (BASELINE.json configs[3])")
    else:
        S = a.num_steps_per_obs or 400
        noisy = a.config == "fhn_noisy"
        wl = FhnWorkload(B, num_steps_per_obs=S, num_obs=a.num_obs, sigma=0.1 if noisy else None, device=devno,
                         chain_offset=rank * B, total_chains=world * B, use_gaussian_splitting=gaussian,
                         num_steps_per_obs_data=a.data_steps_per_obs)
        step_size = a.step_size if a.step_size is not None else 0.1
        workload = (f"FHN {'noisy' if noisy else 'noiseless'}-obs, T={a.num_obs} obs, S={S} steps/obs, R=5 obs/subseq, "
                    f"{'sigma_y=0.1, ' if noisy else ''}{B} chains per GPU "
                    f"(BASELINE.json configs[{1 if noisy else 2}])")
    wl.solver["newton"] = a.solver == "newton"
    workload += f", {'Newton' if wl.solver['newton'] else 'quasi-Newton'} solver, {a.splitting} splitting"
    burn_step = a.burn_step_size if a.burn_step_size is not None else step_size
    ctx = wl.ctx

    # ---- untimed burn-in: the initial states are far from the typical set
    for _ in range(a.burn_iters):
        wl.refresh_momentum()
        act = np.ones(B, dtype=np.int32)
        for _ in range(a.burn_steps):
            r = wl.step(burn_step, active=act)
            act &= (r["status"] == 0).astype(np.int32)
        ctx.switch_partition()

    def run_steps(n, stats=None):
        """n batched leapfrog steps = n / traj_len trajectories.  Between trajectories: IndependentMomentumTransition +
        SwitchPartitionTransition.  A trajectory is ONE chmc_leapfrog_steps call (each chain takes its traj_len steps at its
        own pace; an integrator error ends that chain's trajectory, as in Mici's integration transitions), or with
        --lockstep one batched chmc_leapfrog_step per step with the same bookkeeping on the host."""
        k = run_steps.k
        done = 0
        while done < n:
            if k % a.traj_len == 0:
                if k:
                    ctx.switch_partition()
                wl.refresh_momentum()  # device-side Philox stream + projection
                run_steps.act = np.ones(B, dtype=np.int32)
            m = min(a.traj_len - k % a.traj_len, n - done)
            if a.lockstep:
                for _ in range(m):
                    r = wl.step(step_size, active=run_steps.act)
                    on = run_steps.act == 1
                    if stats is not None:
                        stats.append(dict(attempted=int(on.sum()), ok=int((r["status"][on] == 0).sum()),
                                          iters=int(r["iters_fwd"][on].sum() + r["iters_bwd"][on].sum()),
                                          iters_ok=int((r["iters_fwd"] + r["iters_bwd"])[on & (r["status"] == 0)].sum()),
                                          steps_clean=int((on & (r["status"] == 0)).sum())))
                    run_steps.act &= (r["status"] == 0).astype(np.int32)
            else:
                r = ctx.leapfrog_steps(step_size, m, active=run_steps.act, **wl.solver)
                on = run_steps.act == 1
                if stats is not None:
                    stats.append(dict(attempted=int((r["n_done"][on] + (r["status"][on] > 0)).sum()), ok=int(r["n_done"][on].sum()),
                                      iters=int(r["iters_fwd"][on].sum() + r["iters_bwd"][on].sum()),
                                      iters_ok=int((r["iters_fwd"] + r["iters_bwd"])[on & (r["status"] == 0)].sum()),
                                      steps_clean=int(r["n_done"][on & (r["status"] == 0)].sum())))
                run_steps.act &= (r["status"] == 0).astype(np.int32)
            k += m
            done += m
        run_steps.k = k

    run_steps.k = 0
    # warm-up steps double as the full per-class profiling pass (HIP events around every launch perturb the host
    # side, so the timed region only records events for the dominant kernel class found here)
    L.chmc_profile_enable(1)
    run_steps(a.warmup)
    ms_w = np.zeros(10)
    nl_w = np.zeros(10, dtype=np.int64)
    L.chmc_profile_get(ms_w.ctypes.data_as(_lib.dp), nl_w.ctypes.data_as(C.POINTER(C.c_longlong)))
    dom = int(np.argmax(ms_w)) if a.warmup > 0 else 1
    D.barrier()
    sync()
    L.chmc_profile_enable(0 if a.no_profile else (1 << dom if dom > 0 else 1))
    L.chmc_profile_stride(a.profile_stride)  # events around every n-th launch of the dominant kernel
    stats = []
    qd = torch.empty((B, ctx.Q), dtype=torch.float64, device=dev)  # gather staging

    cabi_info = None
    if a.collective == "cabi" and not emu:
        # the library's own RCCL communicator: id from rank 0, broadcast as 128 bytes, one chmc_comm_init per rank
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
        if world > 1:
            import torch.distributed as dist
            dist.broadcast(idt, src=0)
        ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
        cabi_info = ctx.comm_info()  # (world, rank) read back from RCCL
        gat = torch.empty((world, B, 7), dtype=torch.float64, device=dev)

    def gather_segment():
        # the single gather of samples of a sampling segment: traced variables per chain (u, v_0, hamiltonian)
        ctx.get_state_device(qd.data_ptr(), None)
        ham = torch.from_numpy(ctx.hamiltonian()[:, :1]).to(dev)
        loc = torch.cat([qd[:, :6], ham], 1).contiguous()
        if cabi_info is not None:
            torch.cuda.synchronize(dev)  # (torch's stream has produced `loc`; the library gathers on its own stream)
            ctx.gather_samples_device(loc.data_ptr(), loc.numel(), gat.data_ptr())
            return gat.reshape(world * B, 7).cpu().numpy() if rank == 0 else None
        return D.gather_samples(loc, equal_shards=True)

    iters_before = ctx.counters()["projection_iterations"]
    gather_segment()  # untimed: first use loads torch's copy / cat kernels and sets up the communicator's buffers
    D.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps(a.steps, stats)
    t_steps = time.perf_counter() - t0
    samples = gather_segment()
    D.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("CHMC_BENCH_VERBOSE"):
        print(f"[rank {rank}] steps {t_steps * 1e3:.1f} ms, gather + sync {(elapsed - t_steps) * 1e3:.1f} ms", file=sys.stderr)
    ms = np.zeros(10)
    nl = np.zeros(10, dtype=np.int64)
    L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), nl.ctypes.data_as(C.POINTER(C.c_longlong)))
    L.chmc_profile_enable(0)
    L.chmc_profile_stride(1)
    counters_after = ctx.counters()["projection_iterations"]
    # run-to-run spread: the same region again (K steps + gather, no HIP events), after the official one
    rep_local = []
    for _ in range(max(a.repeats, 0)):
        st_r = []
        D.barrier()
        sync()
        tr0 = time.perf_counter()
        run_steps(a.steps, st_r)
        gather_segment()
        D.barrier()
        sync()
        rep_local.append((time.perf_counter() - tr0, float(sum(x["attempted"] for x in st_r))))
    t_max = D.max_over_ranks(elapsed)
    per_rank = D.gather_samples(np.array([[elapsed * 1e3]]), equal_shards=True)

    attempted = float(sum(x["attempted"] for x in stats))
    n_ok = float(sum(x["ok"] for x in stats))
    iters_all = float(sum(x["iters"] for x in stats))
    n_clean = float(sum(x["steps_clean"] for x in stats))
    k_mean = (sum(x["iters_ok"] for x in stats) / n_clean) if n_clean else float("nan")  # (steps of calls without a failure)
    agg = D.sum_over_ranks([n_ok, attempted, iters_all, 1.0])
    repeats = None
    if rep_local:
        rt = [D.max_over_ranks(t) for t, _ in rep_local]
        ra = [float(D.sum_over_ranks([n])[0]) for _, n in rep_local]
        vals = [n / t for n, t in zip(ra, rt)]
        repeats = {"values": [round(v, 1) for v in vals], "min": round(min(vals), 1), "max": round(max(vals), 1),
                   "note": "the timed region repeated after the official one (same K steps, no HIP events)"}
    if rank == 0:
        # chain-steps actually attempted: a chain whose step fails ends its trajectory ("failed steps count
        # as work done and end that chain's trajectory"); with every step succeeding this is chains x steps
        total_steps = float(agg[1])
        value = total_steps / t_max
        cm = class_model(ctx, wl, model, wl.solver["newton"])
        # measured HBM bytes per launch of every class (two separate rocprofv3 --pmc passes, tools/pmc_summary.py);
        # only quoted when the file was produced by THIS build of the library
        traffic_all, traffic_note = {}, "profiles/traffic.json absent"
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and not emu:
            try:
                tj = json.load(open(tf))
                key = config_key(a.config, S, B)
                sect = tj.get("configs", {}).get(key)
                switches = sorted(k for k in os.environ if k.startswith("CHMC_") and k not in (
                    "CHMC_BENCH_DEVICE", "CHMC_BENCH_VERBOSE", "CHMC_HIP_LIBRARY", "CHMC_DIST_BACKEND"))
                if tj.get("_lib_sha256") != lib_sha256():
                    traffic_note = "profiles/traffic.json is from another build of the library: not quoted"
                elif switches:  # (the counters were taken with the default kernel choices)
                    traffic_note = f"environment switches {switches} select other kernels than the profiled defaults: not quoted"
                elif wl.solver["newton"] is False or gaussian or sect is None:
                    traffic_note = f"profiles/traffic.json holds no counters for this workload ({key}): not quoted"
                else:
                    traffic_all = sect
                    traffic_note = f"profiles/traffic.json [{key}] ({sect.get('_command', 'rocprofv3 --pmc passes')}), same library build"
            except Exception as e:  # noqa: BLE001
                traffic_note = f"profiles/traffic.json unreadable ({e})"
        # chains one launch processes: the Newton-loop kernels are masked per chain, so the later iterations of a solve
        # (and the one speculatively enqueued iteration that finds every chain converged) touch only the chains still
        # active.  For newton_blk the (chain, iteration) pairs of the timed region are known exactly from the
        # per-chain iteration counters; the other classes run over every chain.
        name = _lib.KERNEL_CLASSES[dom]
        avg_ms = ms[dom] / max(nl[dom], 1) if nl[dom] else float("nan")
        launches_all = nl[dom] * a.profile_stride
        pairs = iters_all  # (chain, Newton iteration) pairs of the timed region
        if name == "newton_blk" and nl[dom]:
            chains_per_launch = pairs / launches_all
        elif name in ("update", "constr", "solve_chain", "sym_blk") and nl[dom]:
            # one masked launch per Newton / quasi-Newton iteration (the chains still iterating) plus the launches
            # that run over every chain (state evaluation, momentum projections)
            full = max(launches_all - float(counters_after - iters_before), 0.0)
            chains_per_launch = (pairs + B * full) / launches_all
        else:
            chains_per_launch = float(B)
        mdl = cm.get(name, dict(bytes=8.0 * 3 * ctx.Q, flops=2.0 * ctx.Q))
        traffic = traffic_all.get(name)
        alg_bytes = mdl["bytes"] * chains_per_launch          # SURVEY.md 8(d) operator-level bytes of the launch's chains
        hbm_bytes = mdl.get("hbm_bytes", mdl["bytes"]) * chains_per_launch  # what the fused kernel has to move
        alg_flops = mdl["flops"] * chains_per_launch
        sec = avg_ms * 1e-3
        # HARDWARE fractions of the dominant class: HBM (counter bytes when profiles/traffic.json belongs to this build,
        # otherwise the bytes the kernel has to move) and fp64 vector flops; `frac` is the larger of the two and `bound`
        # names it.  The builder-measured latency floor of the forward scan is a secondary figure, not `frac`.
        hbm_frac_alg = hbm_bytes / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else None
        hbm_frac_cnt = (traffic / sec / 1e9 / HBM_PEAK_GBS) if (traffic and sec > 0) else None
        fp64_frac = alg_flops / sec / 1e12 / FP64_VALU_PEAK_TFLOPS if sec > 0 else None
        hbm_frac = hbm_frac_cnt if hbm_frac_cnt is not None else hbm_frac_alg
        if (fp64_frac or 0.0) > (hbm_frac or 0.0):
            bound, achieved, peak, unit, frac = "fp64_valu", alg_flops / sec / 1e12, FP64_VALU_PEAK_TFLOPS, "TFLOP/s", fp64_frac
        else:
            bound, achieved, peak, unit, frac = ("hbm", (traffic if hbm_frac_cnt is not None else hbm_bytes) / sec / 1e9,
                                                 HBM_PEAK_GBS, "GB/s", hbm_frac)
        roofline = {
            "bound": bound, "kernel": name, "achieved": achieved, "peak": peak, "unit": unit, "frac": frac, "traffic": traffic,
            "hbm_frac_counters": hbm_frac_cnt, "hbm_frac_bytes_to_move": hbm_frac_alg, "fp64_valu_frac": fp64_frac,
            "traffic_source": traffic_note,
            "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_to_move_per_launch": hbm_bytes,
            "operator_level_byte_rate_GBs_not_a_utilisation": alg_bytes / sec / 1e9 if sec > 0 else None,
            "chains_per_launch": chains_per_launch, "avg_launch_ms": avg_ms, "launches_timed": int(nl[dom]),
            "timed_every": a.profile_stride,
            "note": "frac = the larger HARDWARE fraction of the dominant kernel class: HBM bytes of the launch (rocprofv3 "
                    "FETCH/WRITE counters from profiles/traffic.json when that file was produced by this build, else the "
                    "bytes the fused kernel has to move) / HIP-event launch time / 8 TB/s, or algorithmic fp64 flops "
                    "(SURVEY.md 8d) / time / 78.6 TFLOP/s.  `operator_level_byte_rate...` prices SURVEY.md 8(d)'s unfused "
                    "operator decomposition and may exceed the HBM peak: it is a throughput figure, never a utilisation.",
        }
        if name == "constr":
            # secondary: the sequential recursion's dependent-instruction latency floor (tools/ubench/fwd_latency.hip,
            # output kept in profiles/r03_fwd_latency_ubench.txt: 36 ns per FitzHugh-Nagumo step at one wavefront per SIMD)
            steps = max(b["nsteps"] for b in ctx.blocks[ctx.partition])
            roofline["sequential_steps_per_launch"] = steps
            roofline["Msteps_per_s_per_lane"] = steps / sec / 1e6 if sec > 0 else None
            if model == "fhn":
                roofline["latency_floor_frac"] = steps * 36e-9 / sec if sec > 0 else None
                roofline["latency_floor_source"] = "profiles/r03_fwd_latency_ubench.txt (36 ns per step)"
            roofline["note"] += ("  The forward scans keep B x K lanes = a few dozen wavefronts busy with a sequential "
                                 "nonlinear recursion, so neither HBM nor the vector pipes are the limit; "
                                 "`latency_floor_frac` = steps x 36 ns / launch time is the builder-measured ceiling.")
        # the whole step attributed: every class with its time, binding resource and fraction of that resource's peak
        table = {}
        for i, k in enumerate(_lib.KERNEL_CLASSES):
            if not nl_w[i]:
                continue
            per_launch = ms_w[i] / nl_w[i]
            row = {"ms_per_step": round(ms_w[i] / max(a.warmup, 1), 3), "ms_per_launch": round(per_launch, 4),
                   "launches_per_step": round(nl_w[i] / max(a.warmup, 1), 1), "bound": BOUND.get(k, "latency")}
            if k in cm:
                row["compulsory_bytes_per_chain"] = cm[k]["bytes"]
                if "hbm_bytes" in cm[k]:
                    row["hbm_bytes_per_chain"] = cm[k]["hbm_bytes"]
                if BOUND.get(k) == "hbm" and k != "update":  # (update launches are masked: chains per launch unknown here)
                    row["frac_of_hbm_peak"] = round(cm[k].get("hbm_bytes", cm[k]["bytes"]) * B / (per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
                if BOUND.get(k) == "fp64_valu" and k not in ("newton_blk", "constr"):
                    row["frac_of_fp64_peak"] = round(cm[k]["flops"] * B / (per_launch * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS, 3)
            if traffic_all.get(k):
                row["counter_bytes_per_launch"] = traffic_all[k]
                row["hbm_traffic_frac"] = round(traffic_all[k] / (per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
            table[k] = row
        # whole step against HBM: sum over classes of counter bytes per launch x launches per step / time / 8 TB/s
        whole_bytes = whole_frac = None
        if traffic_all:
            whole_bytes = float(sum(traffic_all[k] * table[k]["launches_per_step"] for k in table if traffic_all.get(k)))
            whole_frac = whole_bytes / (t_max / a.steps) / 1e9 / HBM_PEAK_GBS
        rounds_per_step = (counters_after - iters_before) / max(a.steps, 1)
        out = {
            "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": t_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not emu else "emu-rehearsal (TEST ONLY, no GPU)",
            "config": {
                "workload": workload, "chains_per_gpu": B, "global_chains": world * B, "dim_q": ctx.Q,
                "step_size": step_size, "traj_len": a.traj_len,
                "stepping": ("chmc_leapfrog_step per step, host bookkeeping" if a.lockstep else
                             "chmc_leapfrog_steps per trajectory (batched steps)"),
                "chain_steps_attempted": total_steps, "chain_steps_nominal": world * B * a.steps,
                # (`value` counts chain-steps ATTEMPTED -- a chain whose step fails ends its trajectory, SURVEY 8d;
                # rounds 1-2 quoted chains x steps / time, which is this figure)
                "nominal_steps_per_s": world * B * a.steps / t_max,
                "value_repeats": repeats,
                "parallelism": f"chains x{world} (no data-path collective, 1 gather)",
                "ranks_joined": int(agg[3]), "per_rank_ms": None if per_rank is None else [round(float(x), 2) for x in per_rank[:, 0]],
                "mean_newton_iters_fwd_plus_bwd": k_mean, "step_success_rate": float(agg[0] / agg[1]),
                "gathered_sample_shape": None if samples is None else list(samples.shape),
                "bytes_per_chain_step_algorithmic": wl.bytes_per_chain_step(k_mean, newton=wl.solver["newton"]),
                # SURVEY.md 8(d)'s operator-level bytes x steps/s: NOT a DRAM utilisation (the fused path never moves
                # those bytes; the figure can exceed the 8 TB/s peak)
                "operator_level_byte_rate_GBs_not_a_utilisation":
                    wl.bytes_per_chain_step(k_mean, newton=wl.solver["newton"]) * value / 1e9 / world,
                "whole_step_counter_bytes": whole_bytes, "whole_step_hbm_frac": whole_frac,
                "launches_per_step": round(float(sum(nl_w)) / max(a.warmup, 1), 1),
                "newton_rounds_per_step": rounds_per_step,
                "collective_backend": D.backend_name(), "collective_world_size": D.world_size(),
                "collective_route": ("C ABI: chmc_comm_init + chmc_gather_samples (ncclAllGather), RCCL communicator reports "
                                     f"world {cabi_info[0]}, this rank {cabi_info[1]}") if cabi_info is not None
                                    else "torch.distributed gather",
                "cabi_comm_world_size": None if cabi_info is None else cabi_info[0],
                "overlap_halves": int(os.environ.get("CHMC_HALVES", "-1")),
                "kernel_classes_warmup": table,
            },
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            q, p, xo, part = ctx.get_state()
            n_cpu = a.cpu_steps if a.cpu_steps is not None else (250 if model == "fhn" and S >= 400 else 400)
            out["cpu_baseline"] = cpu_baseline(wl, model, q, p, xo, part, step_size, n_cpu, wl.solver, gaussian)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if a.cpu_autodiff_baseline:
                out["cpu_baseline"]["autodiff_restatement_S50"] = cpu_autodiff_baseline()
            elif a.config == "fhn_noisy" and not a.no_other_configs:  # (the default line: a bounded sample of baseline (ii))
                out["cpu_baseline"]["autodiff_restatement_S50_T20"] = cpu_autodiff_baseline(20)
        if world == 1 and a.config == "fhn_noisy" and not a.no_other_configs and not emu and wl.solver["newton"] \
                and not gaussian and B == 256 and S == 400:
            out["config"]["other_configs"] = run_other_configs(a)
        print(json.dumps(_finite(out)), flush=True)
    D.barrier()
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
