#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as MI355X_MICROARCH.md
prescribes: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`) into
per-launch HBM traffic per kernel class -> profiles/traffic.json (read by bench.py for `roofline.traffic`).

Units / corrections (guide, section HBM): counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a
wide (16 B / lane) coalesced streaming read; other access patterns are calibrated on this library's own patterns
(tools/ubench/fetch_calib.hip -> profiles/fetch_calibration.json: bytes read / FETCH_SIZE per pattern, and which pattern
each kernel class follows); the factor used per class is recorded in the output (2.0 when no calibration file exists).
WRITE_SIZE is taken as is.
The output records the SHA-256 of the library that was profiled (`_lib_sha256`), the bench configuration and the
command, and bench.py quotes `roofline.traffic` from it only when they match the build it is timing.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [workload key] [command]
(the output file is keyed by workload: bench.py config_key -- fhn_noisy, fhn_noiseless, sir, fhn_noisy_s800_b512 -- and is
started afresh when the library build changes)"""
import hashlib
import json
import os
import sys
import pandas as pd

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "manifold_mcmc_for_diffusions_amd",
                  "libchmc_hip.so")

CLASS_OF = [  # substring of the kernel name -> profiling class of include/chmc.h (first match wins)
    ("k_newton_lean<chmc::FhnModel, 7, true, true>", "state_blk"), ("k_newton_lean<chmc::FhnModel, 6, true, true>", "state_blk"),
    ("k_newton_lean<chmc::FhnModel, 8, true, true>", "state_blk"), ("k_newton_lean<chmc::SirModel, 8, true, true>", "state_blk"),
    ("k_newton_lean<chmc::FhnNbModel, 6, true, true>", "state_blk"), ("k_newton_lean<chmc::FhnNbModel, 8, true, true>", "state_blk"),
    ("k_newton_ivl<chmc::SirModel, true>", "state_blk"), ("k_newton_comb<chmc::SirModel, 16, true", "state_blk"),
    ("k_newton_lean", "newton_blk"), ("k_newton_ivl", "newton_blk"), ("k_newton_comb", "newton_blk"), ("k_retract_chain", "newton_blk"), ("k_traj_chain", "newton_blk"),
    ("k_newton_factor_wave", "sym_blk"), ("k_gram_rows", "newton_blk"), ("KUpdatePB", "update"), ("KMuF", "solve_chain"),
    ("k_jw_pb", "jacob_vec"), ("KRowsFromPB", "state_blk"),
    ("k_rev_wave_ldsrows<chmc::SirModel, 16, 1", "newton_blk"), ("k_rev_wave_ldsrows<chmc::SirVsModel, 16, 1", "newton_blk"),
    ("k_rev_wave_ldsrows", "state_blk"),
    ("k_rev_wave<chmc::FhnModel, 7, 1", "newton_blk"), ("k_rev_wave<chmc::FhnModel, 7, 0", "state_blk"),
    ("k_rev_wave<chmc::FhnModel, 6, 1", "newton_blk"), ("k_rev_wave<chmc::FhnModel, 6, 0", "state_blk"),
    ("k_rev_wave<chmc::SirModel, 16, 1", "newton_blk"), ("k_rev_wave<chmc::SirModel, 16, 0", "state_blk"),
    ("k_gld_", "grad_log_det_blk"), ("KGldPrep", "sym_blk"), ("KUpdate", "update"), ("k_solve_chain_wave", "solve_chain"),
    ("k_jw_wave", "jacob_vec"), ("KFwd", "constr"), ("k_fwd_scan", "constr"), ("k_fwd_par", "constr"),
    ("KKick", "elementwise"), ("KFlow", "elementwise"),
    ("KMomFix", "elementwise"), ("KRevDiff", "elementwise"), ("Factor", "sym_blk"), ("KSymBlk", "sym_blk"),
]


def per_class(path, counter):
    df = pd.read_csv(path)
    df = df[df["Counter_Name"] == counter]
    # k_traj_chain (single-block layouts) walks a whole trajectory per launch: the burn-in of bench.py launches it for ONE step
    # at a time, the warm-up and the timed region for traj_len steps.  Only the last two launches (warm-up and timed
    # region of `--steps 16 --warmup 16`) have the shape of the launches bench.py times.
    tr = df[df["Kernel_Name"].str.contains("k_traj_chain")]
    if len(tr) > 2:
        order = "Dispatch_Id" if "Dispatch_Id" in tr.columns else None
        tr = tr.sort_values(order) if order else tr
        df = df.drop(tr.index[:-2])
    out = {}
    for _, r in df.iterrows():
        for sub, cls in CLASS_OF:
            if sub in r["Kernel_Name"]:
                s, n = out.get(cls, (0.0, 0))
                out[cls] = (s + float(r["Counter_Value"]), n + 1)
                break
    return out


def fetch_factors():
    """FETCH_SIZE correction per kernel class from the calibration run (profiles/fetch_calibration.json)."""
    path = os.path.join(os.path.dirname(SO), "..", "profiles", "fetch_calibration.json")
    try:
        cal = json.load(open(path))
    except (OSError, ValueError):
        return {}, "no calibration file: factor 2.0 (the guide's 16-B-per-lane figure) for every class"
    fac = {}
    for cls, pat in cal.get("class_pattern", {}).items():
        v = cal.get("factor", {}).get(pat)
        if v:
            fac[cls] = float(v)
    return fac, "profiles/fetch_calibration.json (tools/ubench/fetch_calib.hip on this GPU)"


def main():
    f = per_class(sys.argv[1], "FETCH_SIZE")
    w = per_class(sys.argv[2], "WRITE_SIZE")
    fac, fac_src = fetch_factors()
    sha = hashlib.sha256(open(SO, "rb").read()).hexdigest()
    key = sys.argv[4] if len(sys.argv) > 4 else "fhn_noisy"
    # one file for every profiled workload of ONE library build: {"_lib_sha256", "configs": {key: {class: bytes per launch}}}
    res = {}
    if os.path.exists(sys.argv[3]):
        try:
            res = json.load(open(sys.argv[3]))
        except ValueError:
            res = {}
    if res.get("_lib_sha256") != sha or "configs" not in res:
        res = {"_method": "bytes per launch = (factor * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of the "
                          "class; factor = calibrated bytes-per-FETCH_SIZE of the class's access pattern (`_fetch_factor`); "
                          "two separate rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 2` per workload (single-block layouts: `--steps 16 "
                          "--warmup 16`, whole trajectories, and only the last two launches of k_traj_chain count)",
               "_fetch_factor_source": fac_src, "_lib_sha256": sha, "configs": {}}
    sect = {"_fetch_factor": {}}
    sect["_command"] = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2"
    for cls in sorted(set(f) | set(w)):
        fs, fn = f.get(cls, (0.0, 1))
        ws, wn = w.get(cls, (0.0, 1))
        k = fac.get(cls, 2.0)
        sect["_fetch_factor"][cls] = k
        sect[cls] = (k * fs / max(fn, 1) + ws / max(wn, 1)) * 1024.0
        sect[cls + "_detail"] = {"fetch_KB_raw": fs / max(fn, 1), "write_KB": ws / max(wn, 1), "launches": fn}
    res["configs"][key] = sect
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(key, json.dumps({k: v for k, v in sect.items() if not k.endswith("_detail") and not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
