#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: constrained-leapfrog steps/sec (all chains), FHN noisy-obs, 400 sub-steps.

One "step" = one batched ConstrainedLeapfrogIntegrator.step over the chains resident on a GPU (256 per GPU,
BASELINE.json configs[1]); weak scaling over GPUs (chains shard, no data-path collective, one RCCL gather of the
traced samples).  Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement".

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "constrained-leapfrog steps/sec (all chains), FHN noisy-obs 400 sub-steps"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--chains-per-gpu", type=int, default=256)
    ap.add_argument("--num-steps-per-obs", type=int, default=400)
    ap.add_argument("--step-size", type=float, default=0.1)
    ap.add_argument("--traj-len", type=int, default=16, help="leapfrog steps between momentum refreshes")
    ap.add_argument("--burn-iters", type=int, default=5)
    ap.add_argument("--burn-steps", type=int, default=16)
    ap.add_argument("--burn-step-size", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="record no HIP events in the timed region")
    ap.add_argument("--profile-stride", type=int, default=1,
                    help="timed region: HIP events around every n-th launch of the dominant kernel (an event pair "
                         "costs the stream ~30 us, timing every launch slows the region by ~5 %%)")
    ap.add_argument("--cpu-steps", type=int, default=250)
    return ap.parse_args()


def cpu_baseline(wl, q, p, xo, part, dt, n_steps, solver):
    """The C oracle (kind "port": the reference itself cannot run here) on the host cores, one chain per thread
    (ctypes releases the GIL), on a bounded sample of the same workload: `cores` chains x n_steps leapfrog steps from
    the chains' post-burn-in states."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(len(q), avail, 16))  # a one-GPU box has a 16-core CPU share
    osys = c_oracle.OracleSystem("fhn", wl.obs_interval, wl.S, wl.R, wl.y[:, 0], sigma=wl.sigma)
    chains = []
    for c in range(cores):
        ch = c_oracle.OracleChain(osys)
        ch.set(q[c], p[c], xo[c], part)  # state caches evaluated here, outside the timed region
        chains.append(ch)

    def run(ch):
        n = 0
        for _ in range(n_steps):
            ch.step(dt, newton=solver["newton"], ctol=solver["constraint_tol"], ptol=solver["position_tol"],
                    dtol=solver["divergence_tol"], max_iters=solver["max_iters"], rev_tol=solver["reverse_check_tol"])
            n += 1
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(run, chains))
    el = time.perf_counter() - t0
    # one chain alone on one core (the reference takes its op timings pinned to a single core,
    # run_fhn_model_noiseless_obs_experiments.sh:115): a quarter of the steps, continuing from chain 0's state
    n1 = max(n_steps // 4, 8)
    t1 = time.perf_counter()
    for _ in range(n1):
        chains[0].step(dt, newton=solver["newton"], ctol=solver["constraint_tol"], ptol=solver["position_tol"],
                       dtol=solver["divergence_tol"], max_iters=solver["max_iters"], rev_tol=solver["reverse_check_tol"])
    single = n1 / (time.perf_counter() - t1)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": done / el, "unit": "steps/s", "cores": cores, "kind": "port", "cpu_model": model,
            "per_core": done / el / cores, "single_core_alone": single,
            "sample": f"{cores} chains x {n_steps} leapfrog steps from post-burn-in states, C oracle "
                      f"(oracle/c/chmc_oracle.c, gcc -O2), one chain per host thread, {el:.1f} s"}


def main():
    a = parse()
    from manifold_mcmc_for_diffusions_amd import distributed as D
    rank, local_rank, world = D.init_process_group()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    devno = int(os.environ.get("CHMC_BENCH_DEVICE", local_rank))  # override only for single-GPU rehearsals
    torch.cuda.set_device(devno)
    dev = torch.device("cuda", devno)
    from manifold_mcmc_for_diffusions_amd import _lib
    from manifold_mcmc_for_diffusions_amd.workload import FhnWorkload
    L = _lib.lib()
    B = a.chains_per_gpu
    wl = FhnWorkload(B, num_steps_per_obs=a.num_steps_per_obs, device=devno, chain_offset=rank * B,
                     total_chains=world * B)
    ctx = wl.ctx

    # ---- untimed burn-in: the linear-interpolation initial states are far from the typical set
    for _ in range(a.burn_iters):
        wl.refresh_momentum()
        act = np.ones(B, dtype=np.int32)
        for _ in range(a.burn_steps):
            r = wl.step(a.burn_step_size, active=act)
            act &= (r["status"] == 0).astype(np.int32)
        ctx.switch_partition()

    def run_steps(n, stats=None):
        k = run_steps.k
        for _ in range(n):
            if k % a.traj_len == 0:  # IndependentMomentumTransition + SwitchPartitionTransition between trajectories
                if k:
                    ctx.switch_partition()
                wl.refresh_momentum()  # device-side Philox stream + projection
            r = wl.step(a.step_size)
            k += 1
            if stats is not None:
                stats.append(r)
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
    torch.cuda.synchronize()
    L.chmc_profile_enable(0 if a.no_profile else (1 << dom if dom > 0 else 1))
    L.chmc_profile_stride(a.profile_stride)  # events around every n-th launch of the dominant kernel
    stats = []
    qd = torch.empty((B, ctx.Q), dtype=torch.float64, device=dev)  # gather staging

    def gather_segment():
        # the single gather of samples of a sampling segment: traced variables per chain (u, v_0, hamiltonian)
        ctx.get_state_device(qd.data_ptr(), None)
        ham = torch.from_numpy(ctx.hamiltonian()[:, :1]).to(dev)
        return D.gather_samples(torch.cat([qd[:, :6], ham], 1).contiguous(), equal_shards=True)

    gather_segment()  # untimed: first use loads torch's copy / cat kernels and sets up the communicator's buffers
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(a.steps, stats)
    t_steps = time.perf_counter() - t0
    samples = gather_segment()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if os.environ.get("CHMC_BENCH_VERBOSE"):
        print(f"[rank {rank}] steps {t_steps * 1e3:.1f} ms, gather + sync {(elapsed - t_steps) * 1e3:.1f} ms", file=sys.stderr)
    ms = np.zeros(10)
    nl = np.zeros(10, dtype=np.int64)
    L.chmc_profile_get(ms.ctypes.data_as(_lib.dp), nl.ctypes.data_as(C.POINTER(C.c_longlong)))
    L.chmc_profile_enable(0)
    L.chmc_profile_stride(1)
    t_max = D.max_over_ranks(elapsed)

    status = np.stack([s["status"] for s in stats])
    okm = status == 0
    itf = np.stack([s["iters_fwd"] for s in stats])
    itb = np.stack([s["iters_bwd"] for s in stats])
    k_mean = float((itf[okm] + itb[okm]).mean()) if okm.any() else float("nan")
    agg = D.sum_over_ranks([okm.sum(), status.size, itf.sum() + itb.sum()])
    out = None
    if rank == 0:
        total_steps = world * B * a.steps
        value = total_steps / t_max
        # dominant kernel (largest accumulated device time) and its roofline figure
        name = _lib.KERNEL_CLASSES[dom]
        avg_ms = ms[dom] / max(nl[dom], 1) if nl[dom] else float("nan")
        nnz, Q = wl.nnz(), ctx.Q
        # algorithmic bytes per chain of one launch of each block kernel (SURVEY.md 8d operator decomposition):
        #   newton_blk = constr (Q) + jacob_constr_blocks (Q + nnz written) + lu_jacob_product_blocks (2 nnz)
        #   state_blk  = constr (Q) + jacob (Q + nnz) + chol_gram_blocks (nnz)
        #   grad_log_det_blk = 4 nnz + 2 Q;  update = nnz + 6 Q;  jacob_vec = nnz + Q
        per_chain = {"newton_blk": 8.0 * (3 * nnz + 2 * Q), "state_blk": 8.0 * (2 * nnz + 2 * Q),
                     "grad_log_det_blk": 8.0 * (4 * nnz + 2 * Q), "update": 8.0 * (nnz + 6 * Q),
                     "jacob_vec": 8.0 * (nnz + Q)}.get(name, 8.0 * 3 * Q)
        # chains one launch processes: the Newton-loop kernels are masked per chain, so the later iterations of a solve
        # (and the one speculatively enqueued iteration that finds every chain converged) touch only the chains still
        # active.  For newton_blk the (chain, iteration) pairs of the timed region are known exactly from the
        # per-chain iteration counters; the other classes run over every chain.
        launches_all = nl[dom] * a.profile_stride
        if name == "newton_blk" and nl[dom]:
            chains_per_launch = float(itf.sum() + itb.sum()) / launches_all
        else:
            chains_per_launch = float(B)
        achieved = per_chain * chains_per_launch / (avg_ms * 1e-3) / 1e9 if nl[dom] else float("nan")
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(name)
            except Exception:
                traffic = None
        out = {
            "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": t_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"FHN noisy-obs, T=100 obs, S={a.num_steps_per_obs} steps/obs, R=5 obs/subseq, sigma_y=0.1, "
                            f"{B} chains per GPU (BASELINE.json configs[1]), Newton solver, standard splitting",
                "chains_per_gpu": B, "global_chains": world * B, "dim_q": Q, "step_size": a.step_size,
                "traj_len": a.traj_len, "parallelism": f"chains x{world} (no data-path collective, 1 gather)",
                "mean_newton_iters_fwd_plus_bwd": k_mean, "step_success_rate": float(agg[0] / agg[1]),
                "gathered_sample_shape": None if samples is None else list(samples.shape),
                "bytes_per_chain_step_algorithmic": wl.bytes_per_chain_step(k_mean),
                "whole_path_effective_GBs": wl.bytes_per_chain_step(k_mean) * value / 1e9,
                "warmup_kernel_ms_per_launch": {k: round(ms_w[i] / nl_w[i], 4) for i, k in enumerate(_lib.KERNEL_CLASSES) if nl_w[i]},
                "warmup_kernel_ms_per_step": {k: round(ms_w[i] / max(a.warmup, 1), 3) for i, k in enumerate(_lib.KERNEL_CLASSES) if nl_w[i]},
            },
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_chain * chains_per_launch,
                         "algorithmic_bytes_per_chain": per_chain, "chains_per_launch": chains_per_launch,
                         "avg_launch_ms": avg_ms,
                         "launches_timed": int(nl[dom]), "timed_every": a.profile_stride,
                         "hbm_traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "note": "achieved = operator-level algorithmic bytes (SURVEY.md 8d) / launch time; the kernel "
                                 "never materialises the iterate's Jacobian, so this can exceed the HBM peak; "
                                 "`traffic` is the measured HBM bytes per launch (PMC), `hbm_traffic_frac` its rate "
                                 "against the peak (the kernel is bound by fp64 VALU issue, DESIGN.md section 4)"},
        }
        if world == 1 and not a.no_cpu_baseline:
            q, p, xo, part = ctx.get_state()
            out["cpu_baseline"] = cpu_baseline(wl, q, p, xo, part, a.step_size, a.cpu_steps, wl.solver)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    D.barrier()
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
