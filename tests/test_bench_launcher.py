"""bench.py's own launcher and multi-rank path, rehearsed on CPU: `--gpus 2` without a torchrun environment must start
two ranks itself (torch.distributed.run, gloo here), every rank must join, and the line must say so.  The library
behind it is the TEST-ONLY host emulation build (CHMC_BENCH_EMU_LIB); the GPU path differs only in the loaded
library and the backend (RCCL)."""
import json
import os
import subprocess
import sys
from test_emu_logic import emu_lib, EMU  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--num-obs", "8", "--num-steps-per-obs", "8", "--chains-per-gpu", "3",
         "--burn-iters", "1", "--burn-steps", "2", "--no-cpu-baseline", "--data-steps-per-obs", "50"]


def run_bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CHMC_BENCH_EMU_LIB=EMU, OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=600)


def last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks(emu_lib):  # noqa: F811
    r = run_bench(["--gpus", "2"] + SMALL)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["ranks_joined"] == 2
    assert line["config"]["global_chains"] == 6 and line["config"]["gathered_sample_shape"] == [6, 7]
    assert len(line["config"]["per_rank_ms"]) == 2
    assert line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["step_success_rate"] == 1.0


def test_gpus_1_is_one_rank(emu_lib):  # noqa: F811
    r = run_bench(["--gpus", "1"] + SMALL)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["ranks_joined"] == 1 and line["config"]["global_chains"] == 3


def test_world_size_mismatch_fails_loudly(emu_lib):  # noqa: F811
    r = run_bench(["--gpus", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stdout + r.stderr)
