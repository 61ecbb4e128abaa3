#!/usr/bin/env python3
"""FETCH_SIZE calibration factors from a rocprofv3 --pmc FETCH_SIZE pass over tools/ubench/fetch_calib.bin.
usage: fetch_calib.py <counter_collection.csv> <out.json>
factor[pattern] = bytes the kernel read / (FETCH_SIZE [KB] * 1024), median over the launches of the pattern."""
import json
import sys
import pandas as pd

BYTES = {"calib_coalesced_16B": 2 << 30, "calib_coalesced_8B": 2 << 30, "calib_coalesced_32B": 2 << 30,
         "calib_lane_stream_16B": 5120 * 2000 * 16}
# which pattern the loads of each profiling class follow
CLASS_PATTERN = {
    "elementwise": "coalesced_16B", "update": "coalesced_16B", "other": "coalesced_16B",
    "constr": "lane_stream_16B",
    "newton_blk": "coalesced_8B", "state_blk": "coalesced_8B", "grad_log_det_blk": "coalesced_8B", "jacob_vec": "coalesced_8B",
    "solve_chain": "coalesced_8B", "sym_blk": "coalesced_8B",
}


def main():
    df = pd.read_csv(sys.argv[1])
    df = df[df["Counter_Name"] == "FETCH_SIZE"]
    out = {"_method": "bytes read / (FETCH_SIZE KB * 1024), median over launches; tools/ubench/fetch_calib.hip on MI355X",
           "class_pattern": CLASS_PATTERN, "factor": {}, "raw": {}}
    for name, nbytes in BYTES.items():
        v = df[df["Kernel_Name"].str.contains(name)]["Counter_Value"].astype(float)
        if not len(v):
            continue
        kb = float(v.median())
        out["factor"][name.replace("calib_", "")] = nbytes / (kb * 1024.0) if kb > 0 else None
        out["raw"][name.replace("calib_", "")] = {"fetch_KB_median": kb, "bytes_read": nbytes, "launches": int(len(v))}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out["factor"], indent=1))


if __name__ == "__main__":
    main()
