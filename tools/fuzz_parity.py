"""Random-configuration parity sweep (test infrastructure: uses the oracle): per-op entry points and fused leapfrog
steps of the loaded library against the C oracle for random (model, T, S, R, noise, splitting, solver, metric).
Configurations whose random state is numerically degenerate in the ORACLE (non-finite log-determinant, SIR log-states
beyond 1e6) are skipped; the per-op tolerance is 1e-8 instead of the tests' 1e-10 because random noiseless configurations
are badly conditioned (cond(C) ~ 1e6), and a step whose Newton iteration count differs by one at the edge of the
convergence tolerance is reported as borderline, not as a failure.
usage: python tools/fuzz_parity.py [n_trials] [seed] [emu|hip] [small|mid]    (emu: the test-only host emulation build;
mid: 40-400 steps per observation, i.e. many 64-step tiles per block in the wave kernels)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from manifold_mcmc_for_diffusions_amd import _lib  # noqa: E402

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3 and sys.argv[3] == "emu":
    _lib._LIB = _lib._bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libchmc_emu.so")))
from helpers import make_case, make_ctx, check_ops_against_oracle, check_steps_against_oracle, random_metric  # noqa: E402

mid = len(sys.argv) > 4 and sys.argv[4] == "mid"
rng = np.random.default_rng(seed)
fails, ran, skipped, borderline = [], 0, 0, 0
for trial in range(n_trials):
    model = str(rng.choice(["fhn", "sir", "fhn_nb"]))
    T = int(rng.integers(2, 9 if mid else 15))
    S = int(rng.choice([40, 64, 200, 400] if mid else [3, 5, 8, 8, 16, 24]))
    R = [None, 2, 3, 4, 5, 7][int(rng.integers(0, 6))]
    noisy = bool(rng.integers(0, 2)) or model == "sir"
    gaussian = bool(rng.integers(0, 2)) and model != "sir"
    newton = bool(rng.integers(0, 2))
    metric = bool(rng.integers(0, 2)) and not gaussian
    B = int(rng.choice([1, 3, 5]))
    cseed = int(rng.integers(1e6))
    tag = (model, T, S, R, noisy, gaussian, newton, metric, B, cseed)
    try:
        case = make_case(model, T, S, R, noisy, B=B, seed=cseed, gaussian=gaussian)
        ctx = make_ctx(case)
    except (ValueError, RuntimeError):
        skipped += 1
        continue
    osys = case["osys"]
    if np.abs(case["x_obs"]).max() > 1e6 or not all(
            np.isfinite(osys.gram_ops(case["q"][c], case["x_obs"][c], p, want_grad=False)[2])
            for c in range(B) for p in range(osys.num_partition)):
        skipped += 1
        ctx.close()
        continue
    try:
        if metric:
            M0 = random_metric(rng)
            osys.set_metric(M0), ctx.set_metric(M0)
        check_ops_against_oracle(ctx, case, tol=1e-8)
        dts = np.resize(np.array([0.04, -0.04, 0.07, -0.02, 0.05]), B)
        for part in range(ctx.num_partition):
            check_steps_against_oracle(ctx, case, dts, newton=newton, n_steps=2, part=part)
        ran += 1
    except AssertionError as e:
        if "iters_fwd" in str(e):
            borderline += 1
            print("borderline (iteration count)", tag, flush=True)
        else:
            fails.append((tag, str(e)[:200]))
            print("FAIL", tag, str(e)[:200], flush=True)
    finally:
        osys.set_metric(None)
        ctx.close()
print(f"{ran} configurations agree, {skipped} skipped (unsupported or degenerate), {borderline} borderline, {len(fails)} failures")
sys.exit(1 if fails else 0)
