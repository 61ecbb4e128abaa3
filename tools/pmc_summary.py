#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as MI355X_MICROARCH.md
prescribes: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`) into
per-launch HBM traffic per kernel class -> profiles/traffic.json (read by bench.py for `roofline.traffic`).

Units / corrections (guide, section HBM): counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a
wide (16 B / lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.
The output records the SHA-256 of the library that was profiled (`_lib_sha256`), the bench configuration and the
command, and bench.py quotes `roofline.traffic` from it only when they match the build it is timing.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [config] [command]"""
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
    ("k_newton_lean", "newton_blk"), ("k_newton_ivl", "newton_blk"), ("k_newton_comb", "newton_blk"),
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
    out = {}
    for _, r in df.iterrows():
        for sub, cls in CLASS_OF:
            if sub in r["Kernel_Name"]:
                s, n = out.get(cls, (0.0, 0))
                out[cls] = (s + float(r["Counter_Value"]), n + 1)
                break
    return out


def main():
    f = per_class(sys.argv[1], "FETCH_SIZE")
    w = per_class(sys.argv[2], "WRITE_SIZE")
    res = {"_method": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of the "
                      "class; FETCH_SIZE doubled per the gfx950 correction for 16-B-per-lane coalesced reads; two "
                      "separate rocprofv3 --pmc passes of `bench.py --steps 4 --warmup 2`"}
    res["_lib_sha256"] = hashlib.sha256(open(SO, "rb").read()).hexdigest()
    res["_config"] = sys.argv[4] if len(sys.argv) > 4 else "fhn_noisy"
    res["_command"] = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2"
    for cls in sorted(set(f) | set(w)):
        fs, fn = f.get(cls, (0.0, 1))
        ws, wn = w.get(cls, (0.0, 1))
        res[cls] = (2.0 * fs / max(fn, 1) + ws / max(wn, 1)) * 1024.0
        res[cls + "_detail"] = {"fetch_KB_raw": fs / max(fn, 1), "write_KB": ws / max(wn, 1), "launches": fn}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.endswith("_detail") and not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
