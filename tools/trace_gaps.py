"""Per-kernel busy time and idle gaps of the last N leapfrog steps in a rocprofv3 --kernel-trace --memory-copy-trace run.
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [nsteps]"""
import sys, glob
import pandas as pd

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = pd.read_csv(glob.glob(d + "/*/*kernel_trace.csv")[0])
mc = glob.glob(d + "/*/*memory_copy_trace.csv")
k["name"] = k["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 60)
ev = k[["Start_Timestamp", "End_Timestamp", "name"]]
if mc:
    m = pd.read_csv(mc[0])
    m["name"] = "memcpy:" + m["Direction"].astype(str)
    ev = pd.concat([ev, m[["Start_Timestamp", "End_Timestamp", "name"]]])
ev = ev.sort_values("Start_Timestamp").reset_index(drop=True)
idx = ev.index[ev["name"].str.contains(r"KKickFlowPg|KKick>", regex=True)].tolist()  # first kernel of a leapfrog step
s = idx[-nsteps]
e = ev.index[ev["name"].str.contains("KCommit")].tolist()[-1]
sub = ev.iloc[s:e + 1].copy()
span = (sub["End_Timestamp"].max() - sub["Start_Timestamp"].min()) / 1e6
sub["dur"] = sub["End_Timestamp"] - sub["Start_Timestamp"]
sub["gap"] = (sub["Start_Timestamp"] - sub["End_Timestamp"].cummax().shift(1)).clip(lower=0)
g = sub.groupby("name").agg(n=("dur", "size"), dur_ms=("dur", lambda x: x.sum() / 1e6), gap_ms=("gap", lambda x: x.sum() / 1e6))
g["per_step_ms"] = g["dur_ms"] / nsteps
g["avg_us"] = g["dur_ms"] / g["n"] * 1e3
pd.set_option("display.width", 250)
print(f"steps {nsteps}: span {span / nsteps:.3f} ms/step, busy {sub['dur'].sum() / 1e6 / nsteps:.3f} ms/step, "
      f"gaps {sub['gap'].sum() / 1e6 / nsteps:.3f} ms/step, events/step {len(sub) / nsteps:.0f}")
print(g.sort_values("dur_ms", ascending=False).round(3).to_string())
