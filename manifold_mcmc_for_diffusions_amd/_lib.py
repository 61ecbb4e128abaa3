"""ctypes binding of libchmc_hip.so (C ABI: include/chmc.h).

There is deliberately no fallback: if the HIP library has not been built, or no MI355X is visible when a
context is created, this module raises.  Build with `python __graft_entry__.py` (or `build()` below).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# (CHMC_HIP_LIBRARY: another build of the SAME HIP library, for A/B measurements of kernel variants)
_SO = os.environ.get("CHMC_HIP_LIBRARY") or os.path.join(_PKG, "libchmc_hip.so")
_SRC = os.path.join(_PKG, "csrc")
_LIB = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class ChmcConfig(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("num_obs", C.c_int), ("num_steps_per_obs", C.c_int), ("num_obs_per_subseq", C.c_int),
        ("noisy", C.c_int), ("use_gaussian_splitting", C.c_int), ("num_chains", C.c_int), ("device", C.c_int),
        ("obs_interval", C.c_double), ("sigma", C.c_double), ("y_seq", dp),
    ]


# every symbol include/chmc.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("chmc_create", C.c_int, [C.POINTER(ChmcConfig), C.POINTER(C.c_void_p)]),
    ("chmc_destroy", None, [C.c_void_p]),
    ("chmc_last_error", C.c_char_p, []),
    ("chmc_backend", C.c_char_p, []),
    ("chmc_get_dims", C.c_int, [C.c_void_p, ip]),
    ("chmc_get_blocks", C.c_int, [C.c_void_p, C.c_int, ip]),
    ("chmc_set_state", C.c_int, [C.c_void_p, dp, dp, dp, C.c_int]),
    ("chmc_get_state", C.c_int, [C.c_void_p, dp, dp, dp, ip]),
    ("chmc_init_linear_interpolation", C.c_int, [C.c_void_p, dp, dp, dp, C.c_int]),
    ("chmc_set_metric", C.c_int, [C.c_void_p, dp]),
    ("chmc_tree_leaf", C.c_int, [C.c_void_p, ip, ip, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, dp]),
    ("chmc_tree_begin", C.c_int, [C.c_void_p, dp]),
    ("chmc_tree_set_alive", C.c_int, [C.c_void_p, ip]),
    ("chmc_tree_subtree", C.c_int, [C.c_void_p]),
    ("chmc_tree_step", C.c_int, [C.c_void_p, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                                 dp, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, ip]),
    ("chmc_tree_get", C.c_int, [C.c_void_p, ip, ip, ip, ip, ip, dp, dp]),
    ("chmc_tree_doubling_begin", C.c_int, [C.c_void_p, dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ip]),
    ("chmc_tree_doubling_end", C.c_int, [C.c_void_p, dp, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, ip]),
    ("chmc_tree_get_doubling", C.c_int, [C.c_void_p, ip, ip, dp]),
    ("chmc_set_momentum", C.c_int, [C.c_void_p, dp]),
    ("chmc_get_state_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("chmc_set_momentum_device", C.c_int, [C.c_void_p, C.c_void_p]),
    ("chmc_sample_momentum", C.c_int, [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int]),
    ("chmc_snapshot", C.c_int, [C.c_void_p]),
    ("chmc_restore", C.c_int, [C.c_void_p, ip]),
    ("chmc_restore_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    ("chmc_get_head", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("chmc_update_x_obs_seq", C.c_int, [C.c_void_p]),
    ("chmc_switch_partition", C.c_int, [C.c_void_p]),
    ("chmc_constr", C.c_int, [C.c_void_p, dp]),
    ("chmc_jacob_constr_blocks", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_chol_gram_blocks", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_log_det_sqrt_gram", C.c_int, [C.c_void_p, dp]),
    ("chmc_grad_log_det_sqrt_gram", C.c_int, [C.c_void_p, dp]),
    ("chmc_lmult_by_jacob_constr", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_rmult_by_jacob_constr", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_lmult_by_inv_gram", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_normal_space_component", C.c_int, [C.c_void_p, dp, dp]),
    ("chmc_project_onto_cotangent_space", C.c_int, [C.c_void_p]),
    ("chmc_neg_log_dens_and_grad", C.c_int, [C.c_void_p, dp, C.c_int, dp, dp]),
    ("chmc_neg_log_dens_and_grad_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, dp, C.c_void_p]),
    ("chmc_adam_objective_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, dp]),
    ("chmc_adam_update_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dp, C.c_double,
                                          C.c_double, C.c_double]),
    ("chmc_hamiltonian", C.c_int, [C.c_void_p, dp]),
    ("chmc_project", C.c_int, [C.c_void_p, C.c_int, dp, dp, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, ip,
                               dp, dp, ip]),
    ("chmc_leapfrog_step", C.c_int, [C.c_void_p, dp, ip, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                     C.c_int, C.c_double, ip, ip, ip, dp]),
    ("chmc_leapfrog_steps", C.c_int, [C.c_void_p, dp, ip, ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_double, C.c_int, C.c_double, ip, ip, ip, ip, dp]),
    ("chmc_get_counters", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    ("chmc_get_diagnostics", C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    ("chmc_comm_unique_id", C.c_int, [C.c_void_p]),
    ("chmc_comm_init", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("chmc_gather_samples", C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    ("chmc_comm_info", C.c_int, [C.c_void_p, ip, ip]),
    ("chmc_comm_destroy", C.c_int, [C.c_void_p]),
    ("chmc_profile_enable", C.c_int, [C.c_int]),
    ("chmc_profile_stride", C.c_int, [C.c_int]),
    ("chmc_profile_get", C.c_int, [dp, C.POINTER(C.c_longlong)]),
]
KERNEL_CLASSES = ("other", "newton_blk", "state_blk", "grad_log_det_blk", "update", "solve_chain", "jacob_vec",
                  "constr", "elementwise", "sym_blk")


def build(verbose=False):
    """Compile csrc/chmc.hip for gfx950 into libchmc_hip.so next to this file (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_SRC, f) for f in os.listdir(_SRC)]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", _SO,
           os.path.join(_SRC, "chmc.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_SRC)
    return _SO


def _bind(cdll):
    for name, res, args in SYMBOLS:
        f = getattr(cdll, name)  # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    return cdll


def lib():
    """The loaded HIP library; raises if it is missing (no CPU path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"{_SO} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        # PyTorch-ROCm ships its own HIP runtime; if this library pulled in the system one first, a later
        # torch.cuda initialisation in the same process finds no devices.  Whoever uses torch next to the library
        # (bench.py's gather, the dynamic transition's tree vectors) therefore gets torch's runtime loaded first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _LIB = _bind(C.CDLL(_SO))
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().chmc_last_error()
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc}")


def as_c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(ip)
