"""Thin NumPy-facing wrapper of one `chmc_ctx` (include/chmc.h): a batch of B chains resident on one MI355X.

This is the level bench.py and the multi-GPU driver work at; the Mici-style classes in system.py /
integrators.py sit on top of it.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import ChmcConfig, as_c, ptr, iptr, check

MODEL_IDS = {"fhn": 0, "sir": 1, "fhn_nb": 2}
STATUS_NAMES = {0: "ok", 1: "not_converged", 2: "diverged", 3: "non_reversible", -1: "inactive"}


class ChmcContext:
    def __init__(self, model, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, sigma=None,
                 use_gaussian_splitting=False, num_chains=1, device=0):
        L = _lib.lib()
        self.L = L
        y = as_c(np.asarray(y_seq).reshape(-1))
        self._y = y
        cfg = ChmcConfig(
            model=MODEL_IDS[model] if isinstance(model, str) else int(model), num_obs=len(y),
            num_steps_per_obs=int(num_steps_per_obs),
            num_obs_per_subseq=0 if num_obs_per_subseq is None else int(num_obs_per_subseq),
            # sigma: None (noiseless observations), a number (fixed noise), or "variable": the model's
            # generate_σ_y(u) = exp(u[dim_z]) (sde/example_models/fhn.py:46-47, sir.py:92-93), dim_u = dim_z + 1
            noisy=2 if isinstance(sigma, str) else int(sigma is not None),
            use_gaussian_splitting=int(bool(use_gaussian_splitting)),
            num_chains=int(num_chains), device=int(device), obs_interval=float(obs_interval),
            sigma=0.0 if sigma is None or isinstance(sigma, str) else float(sigma), y_seq=ptr(y))
        h = C.c_void_p()
        check(L.chmc_create(C.byref(cfg), C.byref(h)), "chmc_create")
        self.h = h
        d = np.zeros(16, dtype=np.int32)
        check(L.chmc_get_dims(h, iptr(d)), "chmc_get_dims")
        (self.B, self.Q, self.NV, self.U, self.X, self.T, self.S, self.num_partition, self.RM, self.Kmax) = map(int, d[:10])
        self.M_0 = None  # block metric (set_metric); None = identity
        self.C = [int(d[10]), int(d[11])][: self.num_partition]
        self.K = [int(d[12]), int(d[13])][: self.num_partition]
        self.V, self.V0 = int(d[14]), int(d[15])
        self.noisy = sigma is not None
        self.variable_sigma = isinstance(sigma, str)
        self.sigma = sigma
        self.device = int(device)
        self.partition = 0
        self.blocks = []
        keys = ("obs0", "nobs", "first", "last", "row0", "nrows", "ny", "col0", "ncols", "step0", "nsteps", "_")
        for p in range(self.num_partition):
            raw = np.zeros(self.K[p] * 12, dtype=np.int32)
            check(L.chmc_get_blocks(h, p, iptr(raw)), "chmc_get_blocks")
            self.blocks.append([dict(zip(keys, map(int, raw[12 * b:12 * b + 12]))) for b in range(self.K[p])])

    def close(self):
        if getattr(self, "h", None):
            self.L.chmc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state
    def _bq(self, a, name):
        a = as_c(a)
        if a.shape != (self.B, self.Q):
            raise ValueError(f"{name} must have shape ({self.B}, {self.Q}), got {a.shape}")
        return a

    def set_state(self, q, p, x_obs_seq, partition=0):
        q = self._bq(q, "q")
        p = None if p is None else self._bq(p, "p")
        xo = as_c(x_obs_seq)
        if xo.shape != (self.B, self.T, self.X):
            raise ValueError(f"x_obs_seq must have shape ({self.B}, {self.T}, {self.X})")
        check(self.L.chmc_set_state(self.h, ptr(q), ptr(p), ptr(xo), int(partition)), "chmc_set_state")
        self.partition = int(partition)

    def init_by_linear_interpolation(self, u, v_0, x_obs_seq_init, partition=0):
        """find_initial_state_by_linear_interpolation (sde/mici_extensions.py:1479-1547) for every chain, on the
        device: u [B, U], v_0 [B, V0], x_obs_seq_init [B, T, X]."""
        u, v_0, xo = as_c(u), as_c(v_0), as_c(x_obs_seq_init)
        if u.shape != (self.B, self.U) or v_0.shape != (self.B, self.V0) or xo.shape != (self.B, self.T, self.X):
            raise ValueError(f"expected u ({self.B}, {self.U}), v_0 ({self.B}, {self.V0}), "
                             f"x_obs_seq_init ({self.B}, {self.T}, {self.X})")
        check(self.L.chmc_init_linear_interpolation(self.h, ptr(u), ptr(v_0), ptr(xo), int(partition)),
              "chmc_init_linear_interpolation")
        self.partition = int(partition)

    def get_state(self, want_p=True, want_x_obs=True):
        q = np.empty((self.B, self.Q))
        p = np.empty((self.B, self.Q)) if want_p else None
        xo = np.empty((self.B, self.T, self.X)) if want_x_obs else None
        part = C.c_int(0)
        check(self.L.chmc_get_state(self.h, ptr(q), ptr(p), ptr(xo), C.byref(part)), "chmc_get_state")
        return q, p, xo, part.value

    def set_metric(self, M_0):
        """metric = blockdiag(M_0 [U, U] dense positive definite, identity) (sde/mici_extensions.py:279-315); None = identity.
        The cached factors of the current state are refreshed; the momentum is left as it is."""
        if M_0 is None:
            check(self.L.chmc_set_metric(self.h, None), "chmc_set_metric")
            self.M_0 = None
            return
        m = np.ascontiguousarray(M_0, dtype=np.float64)
        if m.shape != (self.U, self.U):
            raise ValueError(f"M_0 must have shape ({self.U}, {self.U})")
        check(self.L.chmc_set_metric(self.h, ptr(m)), "chmc_set_metric")
        self.M_0 = m.copy()

    def tree_leaf(self, run, take, sub_prop_q_ptr, sub_sum_ptr, ck_p_ptr, ck_sum_ptr, store_slot, check_lo, n_check,
                  ck_end_ptr=None):
        """Fused bookkeeping of one leaf of the batched dynamic-integration trees (chmc_tree_leaf, include/chmc.h):
        returns [B, n_check, 6] = per checked span the two no-U-turn criterion values and, with `ck_end_ptr`, the four
        values of Mici's additional sub-tree checks (zeros where not computed)."""
        out = np.zeros((self.B, n_check, 6))
        r = np.ascontiguousarray(run, dtype=np.int32)
        t = np.ascontiguousarray(take, dtype=np.int32)
        check(self.L.chmc_tree_leaf(self.h, iptr(r), iptr(t), C.c_void_p(sub_prop_q_ptr), C.c_void_p(sub_sum_ptr),
                                    C.c_void_p(ck_p_ptr), C.c_void_p(ck_sum_ptr), C.c_void_p(ck_end_ptr or 0),
                                    int(store_slot), int(check_lo), int(n_check), ptr(out) if n_check else None),
              "chmc_tree_leaf")
        return out

    # ---- per-chain tree decisions on the device (chmc_tree_begin / _subtree / _step / _get, include/chmc.h)
    def tree_begin(self):
        h0 = np.empty(self.B)
        check(self.L.chmc_tree_begin(self.h, ptr(h0)), "chmc_tree_begin")
        return h0

    def tree_set_alive(self, alive):
        a = np.ascontiguousarray(alive, dtype=np.int32)
        check(self.L.chmc_tree_set_alive(self.h, iptr(a)), "chmc_tree_set_alive")

    def tree_subtree(self):
        check(self.L.chmc_tree_subtree(self.h), "chmc_tree_subtree")

    def tree_step(self, dt, u_leaf, max_delta_h, sub_prop_q_ptr, sub_sum_ptr, ck_p_ptr, ck_sum_ptr, ck_end_ptr, store_slot,
                  check_lo, n_check, n_inner_step=1, newton=True, constraint_tol=1e-9, position_tol=1e-8,
                  divergence_tol=1e10, max_iters=50, reverse_check_tol=2e-8):
        """One leaf for every chain still running in its sub-tree: integrator step, termination tests, multinomial
        weight / proposal update, momentum sums, checkpoints and no-U-turn checks, all on the device.  Returns the
        number of chains still running."""
        dt = as_c(np.broadcast_to(np.asarray(dt, dtype=np.float64), (self.B,)))
        u = as_c(np.asarray(u_leaf, dtype=np.float64))
        n = C.c_int(0)
        check(self.L.chmc_tree_step(self.h, ptr(dt), int(n_inner_step), int(newton), constraint_tol, position_tol,
                                    divergence_tol, int(max_iters), reverse_check_tol, ptr(u), float(max_delta_h),
                                    C.c_void_p(sub_prop_q_ptr), C.c_void_p(sub_sum_ptr), C.c_void_p(ck_p_ptr),
                                    C.c_void_p(ck_sum_ptr), C.c_void_p(ck_end_ptr or 0), int(store_slot), int(check_lo),
                                    int(n_check), C.byref(n)), "chmc_tree_step")
        return n.value

    def tree_get(self):
        B = self.B
        o = {k: np.zeros(B, dtype=np.int32) for k in ("alive", "run", "n_step", "failed", "diverged")}
        o["sub_logw"], o["sum_acc"] = np.zeros(B), np.zeros(B)
        check(self.L.chmc_tree_get(self.h, iptr(o["alive"]), iptr(o["run"]), iptr(o["n_step"]), iptr(o["failed"]),
                                   iptr(o["diverged"]), ptr(o["sub_logw"]), ptr(o["sum_acc"])), "chmc_tree_get")
        return o

    def tree_doubling_begin(self, u_dir, neg_q_ptr, neg_p_ptr, pos_q_ptr, pos_p_ptr, sub_sum_ptr):
        """Direction, edge switch (with cache re-evaluation) and sub-tree start of one doubling, on the device; returns the
        number of chains whose tree can still grow."""
        u = as_c(np.asarray(u_dir, dtype=np.float64))
        n = C.c_int(0)
        check(self.L.chmc_tree_doubling_begin(self.h, ptr(u), C.c_void_p(neg_q_ptr), C.c_void_p(neg_p_ptr), C.c_void_p(pos_q_ptr),
                                              C.c_void_p(pos_p_ptr), C.c_void_p(sub_sum_ptr), C.byref(n)),
              "chmc_tree_doubling_begin")
        return n.value

    def tree_doubling_end(self, u_accept, depth, prop_q_ptr, sub_prop_q_ptr, sum_mom_ptr, sub_sum_ptr, neg_q_ptr, neg_p_ptr,
                          pos_q_ptr, pos_p_ptr):
        """Biased progressive sampling, momentum sum, new tree edge and whole-tree no-U-turn criterion of one doubling, on
        the device; returns the number of chains whose tree can still grow."""
        u = as_c(np.asarray(u_accept, dtype=np.float64))
        n = C.c_int(0)
        check(self.L.chmc_tree_doubling_end(self.h, ptr(u), int(depth), C.c_void_p(prop_q_ptr), C.c_void_p(sub_prop_q_ptr),
                                            C.c_void_p(sum_mom_ptr), C.c_void_p(sub_sum_ptr), C.c_void_p(neg_q_ptr),
                                            C.c_void_p(neg_p_ptr), C.c_void_p(pos_q_ptr), C.c_void_p(pos_p_ptr), C.byref(n)),
              "chmc_tree_doubling_end")
        return n.value

    def tree_get_doubling(self):
        moved, depth = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        logw = np.zeros(self.B)
        check(self.L.chmc_tree_get_doubling(self.h, iptr(moved), iptr(depth), ptr(logw)), "chmc_tree_get_doubling")
        return dict(moved=moved != 0, depth=depth.astype(np.int64), logw=logw)

    def set_momentum(self, p):
        p = self._bq(p, "p")
        check(self.L.chmc_set_momentum(self.h, ptr(p)), "chmc_set_momentum")

    def get_state_device(self, q_dev_ptr, p_dev_ptr):
        check(self.L.chmc_get_state_device(self.h, C.c_void_p(q_dev_ptr or 0), C.c_void_p(p_dev_ptr or 0)),
              "chmc_get_state_device")

    def set_momentum_device(self, p_dev_ptr):
        check(self.L.chmc_set_momentum_device(self.h, C.c_void_p(p_dev_ptr)), "chmc_set_momentum_device")

    def sample_momentum(self, seed, draw, chain_offset=0):
        """Momentum refresh on the device: N(0, I) from a counter-based generator, projected onto the cotangent space."""
        check(self.L.chmc_sample_momentum(self.h, int(seed), int(draw), int(chain_offset)), "chmc_sample_momentum")

    def snapshot(self):
        check(self.L.chmc_snapshot(self.h), "chmc_snapshot")

    def restore(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.int32)
        check(self.L.chmc_restore(self.h, iptr(m)), "chmc_restore")

    def restore_device(self, q_dev_ptr, p_dev_ptr, mask, momentum_is_tangent=True):
        m = np.ascontiguousarray(mask, dtype=np.int32)
        check(self.L.chmc_restore_device(self.h, C.c_void_p(q_dev_ptr), C.c_void_p(p_dev_ptr), iptr(m),
                                         int(bool(momentum_is_tangent))), "chmc_restore_device")

    def get_head(self, n):
        out = np.empty((self.B, n))
        check(self.L.chmc_get_head(self.h, int(n), ptr(out)), "chmc_get_head")
        return out

    def update_x_obs_seq(self):
        check(self.L.chmc_update_x_obs_seq(self.h), "chmc_update_x_obs_seq")

    def switch_partition(self):
        check(self.L.chmc_switch_partition(self.h), "chmc_switch_partition")
        self.partition = (self.partition + 1) % self.num_partition

    # ---- per-op
    @property
    def dim_c(self):
        return self.C[self.partition]

    @property
    def num_blocks(self):
        return self.K[self.partition]

    def constr(self):
        c = np.empty((self.B, self.dim_c))
        check(self.L.chmc_constr(self.h, ptr(c)), "chmc_constr")
        return c

    def jacob_constr_blocks(self, want_dv=True):
        du = np.empty((self.B, self.dim_c, self.U))
        dv = np.empty((self.B, self.RM, self.NV)) if want_dv else None
        check(self.L.chmc_jacob_constr_blocks(self.h, ptr(du), ptr(dv)), "chmc_jacob_constr_blocks")
        return du, dv

    def chol_gram_blocks(self):
        cC = np.empty((self.B, self.U, self.U))
        cD = np.empty((self.B, self.num_blocks, self.RM, self.RM))
        check(self.L.chmc_chol_gram_blocks(self.h, ptr(cC), ptr(cD)), "chmc_chol_gram_blocks")
        return cC, cD

    def log_det_sqrt_gram(self):
        out = np.empty(self.B)
        check(self.L.chmc_log_det_sqrt_gram(self.h, ptr(out)), "chmc_log_det_sqrt_gram")
        return out

    def grad_log_det_sqrt_gram(self):
        g = np.empty((self.B, self.Q))
        check(self.L.chmc_grad_log_det_sqrt_gram(self.h, ptr(g)), "chmc_grad_log_det_sqrt_gram")
        return g

    def lmult_by_jacob_constr(self, vct):
        vct = self._bq(vct, "vct")
        out = np.empty((self.B, self.dim_c))
        check(self.L.chmc_lmult_by_jacob_constr(self.h, ptr(vct), ptr(out)), "chmc_lmult_by_jacob_constr")
        return out

    def _bc(self, a, name):
        a = as_c(a)
        if a.shape != (self.B, self.dim_c):
            raise ValueError(f"{name} must have shape ({self.B}, {self.dim_c}), got {a.shape}")
        return a

    def rmult_by_jacob_constr(self, lam):
        lam = self._bc(lam, "vct")
        out = np.empty((self.B, self.Q))
        check(self.L.chmc_rmult_by_jacob_constr(self.h, ptr(lam), ptr(out)), "chmc_rmult_by_jacob_constr")
        return out

    def lmult_by_inv_gram(self, vct):
        vct = self._bc(vct, "vct")
        out = np.empty((self.B, self.dim_c))
        check(self.L.chmc_lmult_by_inv_gram(self.h, ptr(vct), ptr(out)), "chmc_lmult_by_inv_gram")
        return out

    def normal_space_component(self, vct):
        vct = self._bq(vct, "vct")
        out = np.empty((self.B, self.Q))
        check(self.L.chmc_normal_space_component(self.h, ptr(vct), ptr(out)), "chmc_normal_space_component")
        return out

    def project_onto_cotangent_space(self):
        check(self.L.chmc_project_onto_cotangent_space(self.h), "chmc_project_onto_cotangent_space")

    def hamiltonian(self):
        h = np.empty((self.B, 3))
        check(self.L.chmc_hamiltonian(self.h, ptr(h)), "chmc_hamiltonian")
        return h

    def neg_log_dens_and_grad(self, q, use_gaussian_splitting=False, want_grad=True):
        """conditioned_diffusion_neg_log_dens_and_grad (sde/mici_extensions.py:82-205) at q [B, U + V0 + T S V]."""
        QH = self.U + self.NV
        q = as_c(q)
        if q.shape != (self.B, QH):
            raise ValueError(f"q must have shape ({self.B}, {QH}), got {q.shape}")
        val = np.empty(self.B)
        g = np.empty((self.B, QH)) if want_grad else None
        check(self.L.chmc_neg_log_dens_and_grad(self.h, ptr(q), int(bool(use_gaussian_splitting)), ptr(val), ptr(g)),
              "chmc_neg_log_dens_and_grad")
        return val, g

    def neg_log_dens_and_grad_device(self, q_dev_ptr, grad_dev_ptr=None, use_gaussian_splitting=False):
        """The same on device buffers (q [B, U + V0 + T S V] and, optionally, the gradient): returns the values [B]."""
        val = np.empty(self.B)
        check(self.L.chmc_neg_log_dens_and_grad_device(self.h, C.c_void_p(q_dev_ptr), int(bool(use_gaussian_splitting)),
                                                       ptr(val), C.c_void_p(grad_dev_ptr or 0)),
              "chmc_neg_log_dens_and_grad_device")
        return val

    def adam_objective_device(self, u_v_dev_ptr, grad_dev_ptr):
        """One read-back per Adam iteration of the initial-state finder: [B, 3] = objective, |u_v|^2, gradient finite."""
        out = np.empty((self.B, 3))
        check(self.L.chmc_adam_objective_device(self.h, C.c_void_p(u_v_dev_ptr), C.c_void_p(grad_dev_ptr), ptr(out)),
              "chmc_adam_objective_device")
        return out

    def adam_update_device(self, u_v_dev_ptr, m_dev_ptr, v_dev_ptr, grad_dev_ptr, coef, b1=0.9, b2=0.999, eps=1e-8):
        """Adam step in place on device buffers; coef [B, 2] = 1 / (1 - b2^t), lr / (1 - b1^t) (0: parameters stay)."""
        coef = as_c(np.asarray(coef, dtype=np.float64).reshape(self.B, 2))
        check(self.L.chmc_adam_update_device(self.h, C.c_void_p(u_v_dev_ptr), C.c_void_p(m_dev_ptr), C.c_void_p(v_dev_ptr),
                                             C.c_void_p(grad_dev_ptr), ptr(coef), float(b1), float(b2), float(eps)),
              "chmc_adam_update_device")

    def project(self, q, dt, newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10, max_iters=50):
        q = self._bq(q, "q")
        dt = as_c(np.broadcast_to(np.asarray(dt, dtype=np.float64), (self.B,)))
        q_out, mu = np.empty((self.B, self.Q)), np.empty((self.B, self.Q))
        iters, status = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        ndq, err = np.empty(self.B), np.empty(self.B)
        check(self.L.chmc_project(self.h, int(newton), ptr(q), ptr(dt), constraint_tol, position_tol, divergence_tol,
                                  int(max_iters), ptr(q_out), ptr(mu), iptr(iters), ptr(ndq), ptr(err), iptr(status)),
              "chmc_project")
        return dict(q=q_out, mu=mu, iters=iters, norm_dq=ndq, err=err, status=status)

    def leapfrog_step(self, dt, active=None, n_inner_step=1, newton=True, constraint_tol=1e-9, position_tol=1e-8,
                      divergence_tol=1e10, max_iters=50, reverse_check_tol=2e-8):
        dt = as_c(np.broadcast_to(np.asarray(dt, dtype=np.float64), (self.B,)))
        act = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
        status = np.zeros(self.B, dtype=np.int32)
        itf, itb = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        rev = np.zeros(self.B)
        check(self.L.chmc_leapfrog_step(self.h, ptr(dt), iptr(act), int(n_inner_step), int(newton), constraint_tol,
                                        position_tol, divergence_tol, int(max_iters), reverse_check_tol, iptr(status),
                                        iptr(itf), iptr(itb), ptr(rev)), "chmc_leapfrog_step")
        return dict(status=status, iters_fwd=itf, iters_bwd=itb, rev_err=rev)

    def leapfrog_steps(self, dt, n_steps, active=None, n_inner_step=1, newton=True, constraint_tol=1e-9, position_tol=1e-8,
                       divergence_tol=1e10, max_iters=50, reverse_check_tol=2e-8):
        """Whole trajectories: up to n_steps (an int, or one int per chain) consecutive integrator steps per chain, every
        chain at its own pace; a chain's trajectory ends at its first failed step (include/chmc.h chmc_leapfrog_steps).
        Returns n_done, status (0, or the failing step's code; -1 inactive), summed iters_fwd / iters_bwd, rev_err."""
        dt = as_c(np.broadcast_to(np.asarray(dt, dtype=np.float64), (self.B,)))
        act = None if active is None else np.ascontiguousarray(active, dtype=np.int32)
        per_chain = None if np.isscalar(n_steps) else np.ascontiguousarray(np.broadcast_to(n_steps, (self.B,)), dtype=np.int32)
        n_all = int(n_steps) if per_chain is None else 0
        ndone, status = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        itf, itb = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        rev = np.zeros(self.B)
        check(self.L.chmc_leapfrog_steps(self.h, ptr(dt), iptr(act), iptr(per_chain), n_all, int(n_inner_step), int(newton),
                                         constraint_tol, position_tol, divergence_tol, int(max_iters), reverse_check_tol,
                                         iptr(ndone), iptr(status), iptr(itf), iptr(itb), ptr(rev)), "chmc_leapfrog_steps")
        return dict(n_done=ndone, status=status, iters_fwd=itf, iters_bwd=itb, rev_err=rev)

    # ---- the one collective of a chain-sharded run, through the library's own RCCL binding (include/chmc.h)
    def comm_init(self, id128, rank, world):
        """Create this context's RCCL communicator (collective: every rank, same 128-byte id from comm_unique_id)."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        check(self.L.chmc_comm_init(self.h, buf, int(rank), int(world)), "chmc_comm_init")
        self.comm_world = int(world)

    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        check(_lib.lib().chmc_comm_unique_id(buf), "chmc_comm_unique_id")
        return bytes(buf)

    def gather_samples_device(self, local_dev_ptr, count, gathered_dev_ptr):
        """All-gather of `count` doubles per rank between device buffers: gathered [world][count], rank-major."""
        check(self.L.chmc_gather_samples(self.h, C.c_void_p(local_dev_ptr), int(count), C.c_void_p(gathered_dev_ptr)),
              "chmc_gather_samples")

    def comm_info(self):
        """(world size, rank) read back from the RCCL communicator."""
        wd, rk = C.c_int(0), C.c_int(0)
        check(self.L.chmc_comm_info(self.h, C.byref(wd), C.byref(rk)), "chmc_comm_info")
        return wd.value, rk.value

    def comm_destroy(self):
        check(self.L.chmc_comm_destroy(self.h), "chmc_comm_destroy")

    def diagnostics(self):
        """Kernel-path diagnostics (include/chmc.h chmc_get_diagnostics): time-parallel scan counters and launches of the
        optional kernel families."""
        out = (C.c_longlong * 80)()
        check(self.L.chmc_get_diagnostics(self.h, out), "chmc_get_diagnostics")
        v = np.array(out[:], dtype=np.int64)
        return dict(par_scan=v[:64], gram_mfma_launches=int(v[64]), gram_valu_launches=int(v[65]),
                    retract_kernel_launches=int(v[66]), traj_kernel_launches=int(v[67]), extra=v[68:80])

    def counters(self):
        out = (C.c_longlong * 8)()
        check(self.L.chmc_get_counters(self.h, out), "chmc_get_counters")
        keys = ("constr", "jacob_constr_blocks", "lu_jacob_product_blocks", "chol_gram_blocks",
                "grad_log_det_sqrt_gram", "leapfrog_step", "projection_iterations", "_")
        return dict(zip(keys, (int(v) for v in out)))
