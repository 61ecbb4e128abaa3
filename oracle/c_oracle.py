"""ORACLE (test infrastructure only): ctypes binding of oracle/c/libchmc_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libchmc_oracle.so")
_lib = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


PORTABLE_CFLAGS = "-O2 -std=c99 -fPIC"          # oracle/c/Makefile: the build every parity test uses (no FMA contraction)
NATIVE_CFLAGS = "-O3 -march=native -std=c99 -fPIC"  # bench.py's cpu_baseline: the strongest single-source CPU build
CFLAGS = PORTABLE_CFLAGS
_libs = {}


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "c", "chmc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def build_native():
    """-O3 -march=native build for the CPU-baseline timing, compiled ON the machine that runs it (keyed by that
    machine's CPU flags, so a library built elsewhere is never loaded)."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        pass
    d = os.path.join(_HERE, "c", "_native")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "libchmc_oracle_%s.so" % hashlib.sha1(flags.encode()).hexdigest()[:12])
    src = os.path.join(_HERE, "c", "chmc_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc"] + NATIVE_CFLAGS.split() + ["-Wno-unused-function", "-Wno-unused-variable", "-shared",
                               "-o", so, src, "-lm"])
    return so


def select_build(kind):
    """"portable" (default; parity tests) or "native" (CPU-baseline timing): which library lib() hands out from now on."""
    global _lib, CFLAGS
    if kind not in ("portable", "native"):
        raise ValueError(kind)
    if kind not in _libs:
        _libs[kind] = _load(build() if kind == "portable" else build_native())
    _lib = _libs[kind]
    CFLAGS = PORTABLE_CFLAGS if kind == "portable" else NATIVE_CFLAGS


def lib():
    if _lib is None:
        select_build("portable")
    return _lib


def _load(path):
    if True:
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_metric.restype = C.c_int
        L.orc_set_metric.argtypes = [C.c_void_p, dp]
        for f in ("orc_dim_q", "orc_num_partition", "orc_rmax", "orc_dim_nv"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_dim_c", "orc_num_blocks"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.orc_block_info.argtypes = [C.c_void_p, C.c_int, C.c_int, ip]
        L.orc_generate_x_obs_seq.argtypes = [C.c_void_p, dp, dp]
        L.orc_constr.argtypes = [C.c_void_p, dp, dp, C.c_int, dp]
        L.orc_jacob_constr_blocks.argtypes = [C.c_void_p, dp, dp, C.c_int, dp, dp, dp]
        L.orc_gram_ops.argtypes = [C.c_void_p, dp, dp, C.c_int, dp, dp, dp, dp]
        L.orc_jacob_products.argtypes = [C.c_void_p, dp, dp, C.c_int, dp, dp, dp, dp, dp, dp]
        L.orc_project.restype = C.c_int
        L.orc_project.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_int, dp, ip, dp, dp]
        L.orc_chain_create.restype = C.c_void_p
        L.orc_chain_create.argtypes = [C.c_void_p]
        L.orc_chain_destroy.argtypes = [C.c_void_p]
        L.orc_chain_set.argtypes = [C.c_void_p, dp, dp, dp, C.c_int]
        L.orc_chain_get.argtypes = [C.c_void_p, dp, dp, dp, ip]
        L.orc_chain_set_mom.argtypes = [C.c_void_p, dp]
        L.orc_chain_project_mom.argtypes = [C.c_void_p]
        L.orc_chain_hamiltonian.restype = C.c_double
        L.orc_chain_hamiltonian.argtypes = [C.c_void_p]
        L.orc_chain_log_det.restype = C.c_double
        L.orc_chain_log_det.argtypes = [C.c_void_p]
        L.orc_chain_switch_partition.argtypes = [C.c_void_p]
        L.orc_chain_step.restype = C.c_int
        L.orc_chain_step.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                     C.c_int, C.c_double, ip, ip, dp]
        L.orc_chain_trace.restype = C.c_int
        L.orc_chain_trace.argtypes = [C.c_void_p, C.c_int, dp, dp]
    return L


def _d(a):
    return a.ctypes.data_as(dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


MODEL_IDS = {"fhn": 0, "sir": 1, "fhn_nb": 2}


class OracleSystem:
    """One conditioned-diffusion system (model + data + discretisation) in the C oracle."""

    def __init__(self, model, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, sigma=None,
                 use_gaussian_splitting=False):
        L = lib()
        y = _c(np.asarray(y_seq).reshape(-1))
        self.T, self.S = len(y), num_steps_per_obs
        self.R = 0 if num_obs_per_subseq is None else num_obs_per_subseq
        self.noisy = sigma is not None
        # sigma: None (noiseless), a number, or "variable": generate_sigma(u) = exp(u[dim_z]), dim_u = dim_z + 1
        self.variable_sigma = isinstance(sigma, str)
        self.h = L.orc_create(MODEL_IDS[model], self.T, self.S, self.R, 2 if self.variable_sigma else int(self.noisy),
                              0.0 if sigma is None or self.variable_sigma else float(sigma),
                              int(use_gaussian_splitting), float(obs_interval), _d(y))
        if not self.h:
            raise ValueError("unsupported configuration")
        self.L = L
        self.Q = L.orc_dim_q(self.h)
        self.num_partition = L.orc_num_partition(self.h)
        self.rmax = L.orc_rmax(self.h)
        self.NV = L.orc_dim_nv(self.h)
        self.X = {"fhn": 2, "sir": 3, "fhn_nb": 2}[model]
        self.U = 4 + int(self.variable_sigma)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def set_metric(self, M0):
        """metric = blockdiag(M0 [U x U] dense positive definite, identity) (sde/mici_extensions.py:303-315); None: identity."""
        rc = self.L.orc_set_metric(self.h, None if M0 is None else _d(_c(np.asarray(M0).reshape(self.U, self.U))))
        if rc == -1:
            raise ValueError("Only identity matrix metric can be used with Gaussian splitting")
        if rc:
            raise ValueError("M0 is not positive definite")
        self.M0 = None if M0 is None else np.array(M0, dtype=np.float64).reshape(self.U, self.U)

    def dim_c(self, p):
        return self.L.orc_dim_c(self.h, p)

    def num_blocks(self, p):
        return self.L.orc_num_blocks(self.h, p)

    def block_info(self, p, b):
        out = np.zeros(10, dtype=np.int32)
        self.L.orc_block_info(self.h, p, b, out.ctypes.data_as(ip))
        keys = ("obs0", "nobs", "row0", "nrows", "ny", "col0", "ncols", "dv_off", "first", "last")
        return dict(zip(keys, (int(v) for v in out)))

    def generate_x_obs_seq(self, q):
        q = _c(q)
        out = np.zeros((self.T, self.X))
        self.L.orc_generate_x_obs_seq(self.h, _d(q), _d(out))
        return out

    def constr(self, q, x_obs, p):
        q, x_obs = _c(q), _c(x_obs)
        c = np.zeros(self.dim_c(p))
        self.L.orc_constr(self.h, _d(q), _d(x_obs), p, _d(c))
        return c

    def jacob_constr_blocks(self, q, x_obs, p):
        """Returns c, dc_du [C,U], dc_dv in row-slot layout [rmax, NV]."""
        q, x_obs = _c(q), _c(x_obs)
        c = np.zeros(self.dim_c(p))
        du = np.zeros((self.dim_c(p), self.U))
        dv = np.zeros((self.rmax, self.NV))
        self.L.orc_jacob_constr_blocks(self.h, _d(q), _d(x_obs), p, _d(c), _d(du), _d(dv))
        return c, du, dv

    def gram_ops(self, q, x_obs, p, want_grad=True):
        q, x_obs = _c(q), _c(x_obs)
        chol_C = np.zeros((self.U, self.U))
        chol_D = np.zeros((self.num_blocks(p), self.rmax, self.rmax))
        ld = C.c_double(0.0)
        grad = np.zeros(self.Q)
        self.L.orc_gram_ops(self.h, _d(q), _d(x_obs), p, _d(chol_C), _d(chol_D), C.byref(ld),
                            _d(grad) if want_grad else None)
        return chol_C, chol_D, ld.value, grad

    def jacob_products(self, q, x_obs, p, w, lam):
        q, x_obs, w, lam = _c(q), _c(x_obs), _c(w), _c(lam)
        Cn = self.dim_c(p)
        Jw, JTl, Gil, nsc = np.zeros(Cn), np.zeros(self.Q), np.zeros(Cn), np.zeros(self.Q)
        self.L.orc_jacob_products(self.h, _d(q), _d(x_obs), p, _d(w), _d(lam), _d(Jw), _d(JTl), _d(Gil), _d(nsc))
        return Jw, JTl, Gil, nsc

    def project(self, newton, q_prev, q, x_obs, p, dt, ctol=1e-9, ptol=1e-8, dtol=1e10, max_iters=50):
        q_prev, q, x_obs = _c(q_prev), _c(q).copy(), _c(x_obs)
        mu = np.zeros(self.Q)
        it, ndq, err = C.c_int(0), C.c_double(0), C.c_double(0)
        st = self.L.orc_project(self.h, int(newton), _d(q_prev), _d(q), _d(x_obs), p, dt, ctol, ptol, dtol, max_iters,
                                _d(mu), C.byref(it), C.byref(ndq), C.byref(err))
        return st, q, mu, it.value, ndq.value, err.value


class OracleChain:
    """A chain state with cached Jacobian / factors / gradient and the leapfrog step."""

    def __init__(self, system):
        self.sys = system
        self.L = system.L
        self.h = self.L.orc_chain_create(system.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_chain_destroy(self.h)
            self.h = None

    def set(self, q, p, x_obs, part):
        q, x_obs = _c(q), _c(x_obs)
        pp = None if p is None else _c(p)
        self.L.orc_chain_set(self.h, _d(q), None if pp is None else _d(pp), _d(x_obs), int(part))

    def get(self):
        s = self.sys
        q, p, xo = np.zeros(s.Q), np.zeros(s.Q), np.zeros((s.T, s.X))
        part = C.c_int(0)
        self.L.orc_chain_get(self.h, _d(q), _d(p), _d(xo), C.byref(part))
        return q, p, xo, part.value

    def set_mom(self, p):
        p = _c(p)
        self.L.orc_chain_set_mom(self.h, _d(p))

    def project_mom(self):
        self.L.orc_chain_project_mom(self.h)

    def hamiltonian(self):
        return self.L.orc_chain_hamiltonian(self.h)

    def log_det(self):
        return self.L.orc_chain_log_det(self.h)

    def switch_partition(self):
        self.L.orc_chain_switch_partition(self.h)

    def step(self, dt, n_inner=1, newton=True, ctol=1e-9, ptol=1e-8, dtol=1e10, max_iters=50, rev_tol=2e-8):
        itf, itb, rev = C.c_int(0), C.c_int(0), C.c_double(0)
        st = self.L.orc_chain_step(self.h, dt, n_inner, int(newton), ctol, ptol, dtol, max_iters, rev_tol,
                                   C.byref(itf), C.byref(itb), C.byref(rev))
        return st, itf.value, itb.value, rev.value

    def trace(self, direction):
        """Per-iteration (|c|_inf of the iterate, |delta q|_inf of its update) of the forward (0) / reverse (1) retraction
        of the last step: entry k holds what the loop condition (sde/mici_extensions.py:1119-1127) saw after k + 1
        iterations."""
        err, ndq = np.zeros(64), np.zeros(64)
        n = self.L.orc_chain_trace(self.h, int(direction), _d(err), _d(ndq))
        return err[:n].copy(), ndq[:n].copy()
