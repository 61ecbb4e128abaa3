"""Overlap analysis of a rocprofv3 --kernel-trace run: for the kernels after the last `skip_frac` of the trace, the span,
the union of busy intervals, the sum of kernel durations (sum > union = kernels of different streams ran concurrently),
per-stream busy time and per-kernel average durations.
usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [tail fraction, default 0.3]"""
import sys, glob
import pandas as pd

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
k = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])
k = k.sort_values("Start_Timestamp").reset_index(drop=True)
k = k.iloc[int(len(k) * (1 - frac)):].copy()
k["name"] = k["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 70)
k["dur"] = k["End_Timestamp"] - k["Start_Timestamp"]
span = k["End_Timestamp"].max() - k["Start_Timestamp"].min()
ev = sorted([(s, 1) for s in k["Start_Timestamp"]] + [(e, -1) for e in k["End_Timestamp"]])
busy = conc = 0
depth, last = 0, ev[0][0]
for t, dlt in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        conc += t - last
    depth += dlt
    last = t
print(f"kernels {len(k)}, span {span / 1e6:.2f} ms, busy(union) {busy / 1e6:.2f} ms, sum of durations {k['dur'].sum() / 1e6:.2f} ms, "
      f">=2 kernels running {conc / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms")
qcol = "Queue_Id" if "Queue_Id" in k.columns else None
scol = "Stream_Id" if "Stream_Id" in k.columns else None
for col in (qcol, scol):
    if col:
        print(col, k.groupby(col)["dur"].agg(["size", "sum"]).assign(ms=lambda x: x["sum"] / 1e6).drop(columns="sum").to_string())
g = k.groupby("name")["dur"].agg(["size", "sum", "mean"]).sort_values("sum", ascending=False)
g["sum_ms"], g["avg_us"] = g["sum"] / 1e6, g["mean"] / 1e3
pd.set_option("display.width", 250)
print(g[["size", "sum_ms", "avg_us"]].round(2).head(25).to_string())
