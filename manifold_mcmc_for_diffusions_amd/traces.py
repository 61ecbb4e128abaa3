"""Trace and summary output of a sampling run, after scripts/utils.py:338-381 and the trace functions
of the experiment scripts (scripts/fhn_model_noisy_obs_chmc_experiment.py:82-99).

The reference lets Mici memory-map one `.npy` file per traced variable and chain and summarises with ArviZ
(neither is available here).  With thousands of device-resident chains the natural unit is the batch: one
memory-mapped `.npy` per variable with the ArviZ axis order [chain, draw, ...], written in place while sampling, and a
`summary.json` holding per-variable mean / sd / split-R-hat / effective sample size plus the run totals the
reference records (`total_sampling_time`, `final_integrator_step_size`, `total_<op>_calls`)."""
import json
import os
import numpy as np


class TraceWriter:
    def __init__(self, directory, n_chain, n_draw, shapes):
        """shapes: {variable name: trailing shape}.  Files `<directory>/trace_<name>.npy`, float64 [chain, draw, ...]."""
        os.makedirs(directory, exist_ok=True)
        self.directory, self.n_chain, self.n_draw = directory, n_chain, n_draw
        self.maps = {k: np.lib.format.open_memmap(os.path.join(directory, f"trace_{k}.npy"), mode="w+",
                                                  dtype=np.float64, shape=(n_chain, n_draw) + tuple(s))
                     for k, s in shapes.items()}

    def write(self, draw, values):
        """values: {name: [chain, ...]} of draw index `draw`."""
        for k, v in values.items():
            self.maps[k][:, draw] = v

    def flush(self):
        for m in self.maps.values():
            m.flush()

    def arrays(self):
        return self.maps


def _autocov(x):
    """Autocovariance of each row of x [m, n] by FFT."""
    n = x.shape[1]
    f = np.fft.rfft(x - x.mean(1, keepdims=True), 2 * n, axis=1)
    return np.fft.irfft(f * np.conj(f), axis=1)[:, :n].real / n


def split_rhat_and_ess(x):
    """Split-R-hat and bulk effective sample size of one scalar variable, x [chain, draw]
    (Gelman et al. BDA3 11.4-11.5; Geyer's initial positive sequence truncation of the combined autocorrelation)."""
    x = np.asarray(x, dtype=np.float64)
    m0, n0 = x.shape
    h = n0 // 2
    if h < 2:
        return float("nan"), float("nan")
    s = np.concatenate([x[:, :h], x[:, h:2 * h]], 0)
    m, n = s.shape
    w = s.var(1, ddof=1).mean()
    b = n * s.mean(1).var(ddof=1) if m > 1 else 0.0
    var_plus = (n - 1) / n * w + b / n
    if not np.isfinite(var_plus) or var_plus <= 0.0:
        return float("nan"), float("nan")
    rhat = float(np.sqrt(var_plus / w)) if w > 0 else float("nan")
    rho = 1.0 - (w - _autocov(s).mean(0) * n / (n - 1)) / var_plus
    rho[0] = 1.0
    tau, t = -1.0, 0
    while t + 1 < n:  # sums of adjacent pairs stay positive
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        tau += 2.0 * pair
        t += 2
    ess = m * n / max(tau, 1.0 / np.log10(max(m * n, 10)))
    return rhat, float(min(ess, m * n * np.log10(max(m * n, 10))))


def summarize(traces, var_names=None):
    """{name or name[i]: {mean, sd, r_hat, ess_bulk}} over [chain, draw, ...] arrays (ArviZ's summary columns)."""
    out = {"mean": {}, "sd": {}, "r_hat": {}, "ess_bulk": {}}
    for k in (var_names or list(traces)):
        a = np.asarray(traces[k])
        flat = a.reshape(a.shape[0], a.shape[1], -1)
        for i in range(flat.shape[2]):
            name = k if flat.shape[2] == 1 else f"{k}[{i}]"
            r, e = split_rhat_and_ess(flat[:, :, i])
            out["mean"][name] = float(flat[:, :, i].mean())
            out["sd"][name] = float(flat[:, :, i].std(ddof=1))
            out["r_hat"][name], out["ess_bulk"][name] = r, e
    return out


def save_summary(directory, traces, var_names, sampling_time, step_size, call_counts=None):
    """summary.json as scripts/utils.py:368-381 writes it."""
    s = summarize(traces, var_names)
    s["total_sampling_time"] = float(sampling_time)
    s["final_integrator_step_size"] = float(step_size)
    for k, v in (call_counts or {}).items():
        s[f"total_{k}_calls"] = int(v)
    with open(os.path.join(directory, "summary.json"), "w") as f:
        json.dump(s, f, ensure_ascii=False, indent=2)
    return s
