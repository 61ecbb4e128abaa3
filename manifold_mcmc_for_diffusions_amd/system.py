"""Mici System / chain-state / projection-solver surface of the reference, backed by the HIP library.

Same names, argument order and error behaviour as `sde.mici_extensions` (reference file:line cited per item);
the differences a C ABI forces are:
  * the four model callables must be the model handles of `example_models` (they select compiled device code);
  * every array may carry a leading chain axis B (`num_chains`); with B == 1 plain reference shapes are accepted
    and returned, and numerical failures raise the reference's exceptions; with B > 1 they are reported per chain;
  * `jacob_constr_blocks` returns the library's layouts (dc_du [C, U], dc_dv row-slot [RM, NV], dc_dn = sigma).
The state an operation refers to is made resident on the device on demand and cached there, which plays the role
of Mici's `cache_in_state` memoisation (:1151-1184).
"""
import itertools
from numbers import Number
import numpy as np
from .context import ChmcContext
from .errors import ConvergenceError, HamiltonianDivergenceError
from .example_models import ModelHandle

_tokens = itertools.count(1)


class IdentityMatrix:
    """Stand-in for mici.matrices.IdentityMatrix."""

    def __init__(self, size=None):
        self.size = size

    def __matmul__(self, other):
        return other

    def __rmatmul__(self, other):
        return other

    @property
    def inv(self):
        return self

    @property
    def sqrt(self):
        return self

    log_abs_det = 0.0


class DensePositiveDefiniteMatrix:
    """Stand-in for mici.matrices.DensePositiveDefiniteMatrix: `.array`, `.inv`, `.sqrt` (lower Cholesky factor),
    `.log_abs_det`, `@` on the last axis of an array."""

    def __init__(self, array):
        self.array = np.array(array, dtype=np.float64)
        if self.array.ndim != 2 or self.array.shape[0] != self.array.shape[1]:
            raise ValueError("array must be square")
        self._chol = np.linalg.cholesky(self.array)  # raises LinAlgError if not positive definite

    @property
    def shape(self):
        return self.array.shape

    def __matmul__(self, other):
        return np.asarray(other) @ self.array.T

    @property
    def inv(self):
        return DensePositiveDefiniteMatrix(np.linalg.inv(self.array))

    @property
    def sqrt(self):
        return _Dense(self._chol)

    @property
    def log_abs_det(self):
        return 2.0 * float(np.log(np.diag(self._chol)).sum())


class _Dense:
    def __init__(self, array):
        self.array = array

    def __matmul__(self, other):
        return np.asarray(other) @ self.array.T


class PositiveDefiniteBlockDiagonalMatrix:
    """Stand-in for mici.matrices.PositiveDefiniteBlockDiagonalMatrix; the system accepts two blocks, a dense
    positive-definite `dim_u x dim_u` one and an identity (sde/mici_extensions.py:279-315)."""

    def __init__(self, blocks):
        self.blocks = tuple(blocks)

    def _apply(self, which, other):
        other = np.asarray(other)
        out, o = [], 0
        for i, b in enumerate(self.blocks):
            n = b.shape[0] if hasattr(b, "shape") else (other.shape[-1] - o if i == len(self.blocks) - 1 else b.size)
            part = other[..., o:o + n]
            out.append(part if _is_identity(b) else getattr(b, which) @ part if which else b @ part)
            o += n
        return np.concatenate(out, -1)

    def __matmul__(self, other):
        return self._apply("", other)

    @property
    def inv(self):
        return _BlockOp(self, "inv")

    @property
    def sqrt(self):
        return _BlockOp(self, "sqrt")

    @property
    def log_abs_det(self):
        return float(sum(b.log_abs_det for b in self.blocks))


class _BlockOp:
    def __init__(self, m, which):
        self.m, self.which = m, which

    def __matmul__(self, other):
        return self.m._apply(self.which, other)


class ScaledIdentity:
    """`scalar * IdentityMatrix()` as returned by dh2_flow_dmom (:1233-1238)."""

    def __init__(self, scalar):
        self.scalar = scalar

    def __matmul__(self, other):
        s = np.asarray(self.scalar)
        return (s[..., None] if s.ndim else s) * other


class _ScaledOp:
    """`dt * metric.inv` (dh2_flow_dmom :1238)."""

    def __init__(self, scalar, op):
        self.scalar, self.op = scalar, op

    def __matmul__(self, other):
        s = np.asarray(self.scalar)
        return (s[..., None] if s.ndim else s) * (self.op @ other)


def _is_identity(metric):
    return metric is None or isinstance(metric, IdentityMatrix) or type(metric).__name__ == "IdentityMatrix"


def _metric_m0(metric, dim_u):
    """M_0 of a supported metric (None for the identity); raises as the reference does (:305-315)."""
    if _is_identity(metric):
        return None
    blocks = getattr(metric, "blocks", None)
    if blocks is not None and len(blocks) == 2 and _is_identity(blocks[1]) and hasattr(blocks[0], "array"):
        m0 = np.asarray(blocks[0].array, dtype=np.float64)
        if m0.shape == (dim_u, dim_u):
            return m0
    raise NotImplementedError("Only identity and block diagonal metrics with identity lower right block currently supported.")


class ConditionedDiffusionHamiltonianState:
    """sde/mici_extensions.py:1285-1320.  Assigning `pos`, `x_obs_seq` or `partition` invalidates everything cached
    for the state (the dependency sets of :1151-1176); copies share the call-count dictionary like Mici's."""

    _deps = ("pos", "x_obs_seq", "partition")

    def __init__(self, pos, x_obs_seq, partition=0, mom=None, dir=1, _call_counts=None, _dependencies=None,
                 _cache=None, _read_only=False):
        object.__setattr__(self, "_call_counts", {} if _call_counts is None else _call_counts)
        object.__setattr__(self, "_token", next(_tokens))
        object.__setattr__(self, "_mom_token", next(_tokens))
        self.pos = np.array(pos, dtype=np.float64)
        self.x_obs_seq = np.array(x_obs_seq, dtype=np.float64)
        self.partition = int(partition)
        self.mom = None if mom is None else np.array(mom, dtype=np.float64)
        self.dir = dir

    def __setattr__(self, k, v):
        if k in self._deps:
            object.__setattr__(self, "_token", next(_tokens))
        elif k == "mom":
            object.__setattr__(self, "_mom_token", next(_tokens))
        object.__setattr__(self, k, v)

    def copy(self):
        new = object.__new__(type(self))
        for k, v in self.__dict__.items():
            object.__setattr__(new, k, v.copy() if isinstance(v, np.ndarray) else v)
        object.__setattr__(new, "_call_counts", self._call_counts)
        return new


class ConditionedDiffusionConstrainedSystem:
    """sde/mici_extensions.py:208-1259."""

    def __init__(self, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq, dim_u, dim_x, dim_v, forward_func,
                 generate_x_0, generate_z, obs_func, generate_σ=None, use_gaussian_splitting=False, metric=None,
                 dim_v_0=None, *, num_chains=1, device=0):
        handles = (forward_func, generate_x_0, generate_z, obs_func)
        if not all(isinstance(h, ModelHandle) for h in handles) or len({h.model for h in handles}) != 1:
            raise TypeError("forward_func, generate_x_0, generate_z and obs_func must be the handles of one model of "
                            "manifold_mcmc_for_diffusions_amd.example_models (device code is selected by model)")
        model = forward_func.model
        if use_gaussian_splitting and not _is_identity(metric):  # :293-300
            raise ValueError("Only identity matrix metric can be used with Gaussian splitting")
        m0 = _metric_m0(metric, dim_u)  # :305-315
        # generate_σ: None (noiseless), a number (:354-358), or the model's generate_σ_y handle: observation noise
        # exp(u[dim_z]) with dim_u = dim_z + 1 (scripts/sir_model_chmc_experiment.py:44,58,77)
        variable_σ = isinstance(generate_σ, ModelHandle)
        if variable_σ and (generate_σ.model is not model or generate_σ.role != "generate_σ_y"):
            raise TypeError("generate_σ must be a number, None, or the generate_σ_y handle of the same model")
        if generate_σ is not None and not variable_σ and not isinstance(generate_σ, Number):
            raise TypeError("generate_σ must be a number, None, or the model's generate_σ_y handle (callables cannot "
                            "cross the C ABI: the device code of generate_σ_y = exp(u[dim_z]) is selected by the handle)")
        y_seq = np.asarray(y_seq, dtype=np.float64)
        if y_seq.ndim != 2 or y_seq.shape[1] != model.dim_y:
            raise ValueError(f"y_seq must have shape (num_obs, {model.dim_y})")
        dim_v_0 = dim_x if dim_v_0 is None else dim_v_0
        if (dim_u, dim_x, dim_v, dim_v_0) != (model.dim_z + int(variable_σ), model.dim_x, model.dim_v, model.dim_v_0):
            raise ValueError("dim_u / dim_x / dim_v / dim_v_0 do not match the model (dim_u = dim_z + 1 with generate_σ_y)")
        self._metric = IdentityMatrix()
        self.use_gaussian_splitting = bool(use_gaussian_splitting)
        self.model = model
        self.ctx = ChmcContext(model.name, obs_interval, num_steps_per_obs, num_obs_per_subseq, y_seq[:, 0],
                               sigma=None if generate_σ is None else "variable" if variable_σ else float(generate_σ),
                               use_gaussian_splitting=use_gaussian_splitting, num_chains=num_chains, device=device)
        self.num_chains = num_chains
        self.num_partition = self.ctx.num_partition  # :362
        δ = obs_interval / num_steps_per_obs
        self.model_dict = {  # :363-377
            "dim_u": dim_u, "dim_v": dim_v, "dim_v_0": dim_v_0, "dim_y": model.dim_y, "num_obs": y_seq.shape[0],
            "num_steps_per_obs": num_steps_per_obs, "δ": δ, "generate_z": generate_z, "generate_x_0": generate_x_0,
            "generate_σ": generate_σ, "forward_func": forward_func, "obs_func": obs_func, "y_seq": y_seq,
        }
        self.y_subseqs = [[y_seq[b["obs0"]:b["obs0"] + b["nobs"]] for b in blocks] for blocks in self.ctx.blocks]
        self._resident = None
        self._resident_mom = None
        if m0 is not None:
            self.metric = metric

    @property
    def metric(self):
        return self._metric

    @metric.setter
    def metric(self, metric):  # assigned by metric adapters at the end of the warm-up (:1926-1931)
        m0 = _metric_m0(metric, self.model_dict["dim_u"])
        if m0 is not None and self.use_gaussian_splitting:
            raise ValueError("Only identity matrix metric can be used with Gaussian splitting")
        self.ctx.set_metric(m0)
        self._metric = IdentityMatrix() if m0 is None else metric
        self._resident = self._resident_mom = None  # the cached factors depend on the metric

    # ---- residency (plays the role of cache_in_state)
    def _b(self, a, tail):
        a = np.asarray(a, dtype=np.float64)
        return a.reshape((self.num_chains,) + tail)

    def _out(self, a, state):
        return a[0] if np.ndim(state.pos) == 1 else a

    def _sync(self, state, mom=False):
        c = self.ctx
        if self._resident != state._token:
            c.set_state(self._b(state.pos, (c.Q,)), None if state.mom is None else self._b(state.mom, (c.Q,)),
                        self._b(state.x_obs_seq, (c.T, c.X)), state.partition)
            self._resident, self._resident_mom = state._token, state._mom_token
            for k in ("jacob_constr_blocks", "chol_gram_blocks", "log_det_sqrt_gram", "grad_log_det_sqrt_gram"):
                state._call_counts[k] = state._call_counts.get(k, 0) + 1
        elif mom and self._resident_mom != state._mom_token and state.mom is not None:
            c.set_momentum(self._b(state.mom, (c.Q,)))
            self._resident_mom = state._mom_token

    def _adopt(self, state):
        """The device now holds `state` (after a device-side update written back into it)."""
        self._resident, self._resident_mom = state._token, state._mom_token

    # ---- System methods
    def constr(self, state):  # :1151-1155
        self._sync(state)
        state._call_counts["constr"] = state._call_counts.get("constr", 0) + 1
        return self._out(self.ctx.constr(), state)

    def jacob_constr_blocks(self, state):  # :1157-1161
        self._sync(state)
        du, dv = self.ctx.jacob_constr_blocks()
        dn = None if self.ctx.sigma is None else self.ctx.sigma
        return self._out(du, state), self._out(dv, state), dn

    def chol_gram_blocks(self, state):  # :1163-1167
        self._sync(state)
        cC, cD = self.ctx.chol_gram_blocks()
        return self._out(cC, state), self._out(cD, state)

    def log_det_sqrt_gram(self, state):  # :1169-1171
        self._sync(state)
        ld = self.ctx.log_det_sqrt_gram()
        return float(ld[0]) if np.ndim(state.pos) == 1 else ld

    def grad_log_det_sqrt_gram(self, state):  # :1173-1184
        self._sync(state)
        return self._out(self.ctx.grad_log_det_sqrt_gram(), state)

    def neg_log_dens(self, state):  # standard_normal_neg_log_dens :56-58
        return 0.5 * np.sum(state.pos ** 2, -1)

    def grad_neg_log_dens(self, state):  # standard_normal_grad_neg_log_dens :61-63
        return state.pos

    def h1(self, state):  # :1186-1190
        if self.use_gaussian_splitting:
            return self.log_det_sqrt_gram(state)
        return self.neg_log_dens(state) + self.log_det_sqrt_gram(state)

    def dh1_dpos(self, state):  # :1192-1196
        if self.use_gaussian_splitting:
            return self.grad_log_det_sqrt_gram(state)
        return self.grad_neg_log_dens(state) + self.grad_log_det_sqrt_gram(state)

    def h2(self, state):  # :1198-1202
        if self.use_gaussian_splitting:
            return 0.5 * np.sum(state.pos ** 2, -1) + 0.5 * np.sum(state.mom ** 2, -1)
        return 0.5 * np.sum(state.mom * (self.metric.inv @ state.mom), -1)

    def h(self, state):
        return self.h1(state) + self.h2(state)

    def dh2_dmom(self, state):  # :1204-1208
        return self.metric.inv @ state.mom

    def dh_dmom(self, state):
        return self.dh2_dmom(state)

    def dh2_dpos(self, state):  # :1210-1214
        return state.pos if self.use_gaussian_splitting else 0 * state.pos

    def dh_dpos(self, state):  # :1216-1220
        if self.use_gaussian_splitting:
            return self.dh1_dpos(state) + self.dh2_dpos(state)
        return self.dh1_dpos(state)

    def h1_flow(self, state, dt):  # mici System.h1_flow
        state.mom = state.mom - np.asarray(dt)[..., None] * self.dh1_dpos(state) if np.ndim(dt) else \
            state.mom - dt * self.dh1_dpos(state)

    def h2_flow(self, state, dt):  # :1222-1231
        dtb = np.asarray(dt)[..., None] if np.ndim(dt) else dt
        if self.use_gaussian_splitting:
            sin_dt, cos_dt = np.sin(dtb), np.cos(dtb)
            pos = state.pos.copy()
            state.pos = state.pos * cos_dt + sin_dt * state.mom
            state.mom = state.mom * cos_dt - sin_dt * pos
        else:
            state.pos = state.pos + dtb * self.dh2_dmom(state)

    def dh2_flow_dmom(self, dt):  # :1233-1238
        if self.use_gaussian_splitting:
            return ScaledIdentity(np.sin(dt)), ScaledIdentity(np.cos(dt))
        return (ScaledIdentity(dt) if _is_identity(self.metric) else _ScaledOp(dt, self.metric.inv)), IdentityMatrix()

    def update_x_obs_seq(self, state):  # :1240-1241
        self._sync(state)
        self.ctx.update_x_obs_seq()
        _, _, xo, _ = self.ctx.get_state(want_p=False)
        state.x_obs_seq = xo[0] if np.ndim(state.pos) == 1 else xo
        # x_obs_seq changed: the device caches (J, factors, gradient) are stale until the next _sync

    def normal_space_component(self, state, vct):  # :1243-1250
        self._sync(state)
        return self._out(self.ctx.normal_space_component(self._b(vct, (self.ctx.Q,))), state)

    def project_onto_cotangent_space(self, mom, state):  # :1252-1254
        mom -= self.normal_space_component(state, mom)
        return mom

    def sample_momentum(self, state, rng):  # :1256-1259
        mom = self.metric.sqrt @ rng.standard_normal(state.pos.shape)
        return self.project_onto_cotangent_space(mom, state)


class SwitchPartitionTransition:
    """sde/mici_extensions.py:1262-1282"""

    state_variables = {"partition", "x_obs_seq"}
    statistic_types = None

    def __init__(self, system):
        self.system = system
        self.num_partition = system.num_partition

    def sample(self, state, rng=None):
        state.partition = (state.partition + 1) % self.num_partition
        self.system.update_x_obs_seq(state)
        return state, None


def _solve_projection(newton, state, state_prev, dt, system, constraint_tol, position_tol, divergence_tol, max_iters):
    c = system.ctx
    system._sync(state_prev)  # J(state_prev) (+ Cholesky factors for quasi-Newton) resident / cached
    res = c.project(system._b(state.pos, (c.Q,)), dt, newton=newton, constraint_tol=constraint_tol,
                    position_tol=position_tol, divergence_tol=divergence_tol, max_iters=max_iters)
    i = res["iters"]
    cc = state._call_counts  # :1382-1387, :1451-1461
    for k in (("constr", "jacob_constr_blocks", "lu_jacob_product_blocks") if newton else ("constr",)):
        cc[k] = cc.get(k, 0) + int(i.sum())
    _, dh2_flow_mom_dmom = system.dh2_flow_dmom(dt)
    ok = res["status"] == 0
    single = np.ndim(state.pos) == 1
    state.last_projection = res
    if ok.all():
        state.pos = res["q"][0] if single else res["q"]
        if state.mom is not None:
            state.mom = state.mom - dh2_flow_mom_dmom @ (res["mu"][0] if single else res["mu"])
        return state
    name = "Newton" if newton else "Quasi-Newton"
    bad = int(np.flatnonzero(~ok)[0])
    err, ndq, it = float(res["err"][bad]), float(res["norm_dq"][bad]), int(i[bad])
    where = "" if single else f" (chain {bad}; statuses {res['status'].tolist()})"
    if res["status"][bad] == 2:  # :1393-1397, :1467-1471
        raise ConvergenceError(f"{name} iteration diverged on iteration {it}. Last |c|={err:.1e}, |δq|={ndq}.{where}")
    raise ConvergenceError(f"{name} iteration did not converge. Last |c|={err:.1e}, |δq|={ndq}.{where}")  # :1398-1402


def jitted_solve_projection_onto_manifold_quasi_newton(state, state_prev, dt, system, constraint_tol=1e-8,
                                                       position_tol=1e-8, divergence_tol=1e10, max_iters=50):
    """sde/mici_extensions.py:1323-1402 (symmetric quasi-Newton retraction)."""
    return _solve_projection(False, state, state_prev, dt, system, constraint_tol, position_tol, divergence_tol,
                             max_iters)


def jitted_solve_projection_onto_manifold_newton(state, state_prev, dt, system, constraint_tol=1e-8,
                                                 position_tol=1e-8, divergence_tol=1e10, max_iters=50):
    """sde/mici_extensions.py:1405-1476 (full Newton retraction)."""
    return _solve_projection(True, state, state_prev, dt, system, constraint_tol, position_tol, divergence_tol,
                             max_iters)


def find_initial_state_by_linear_interpolation(system, rng, generate_x_obs_seq_init, u=None, v_0=None, **model_dict):
    """sde/mici_extensions.py:1479-1547 for one chain (num_chains == 1 systems)."""
    from .init import find_initial_state_by_linear_interpolation as _find
    md = system.model_dict if not model_dict else model_dict
    q, x_obs_seq = _find(system.model, md["δ"] * md["num_steps_per_obs"], md["num_steps_per_obs"], md["y_seq"], rng,
                         generate_x_obs_seq_init, md["generate_σ"] is not None, u=u, v_0=v_0, dim_u=md["dim_u"])
    state = ConditionedDiffusionHamiltonianState(pos=q, x_obs_seq=x_obs_seq)
    state.mom = system.sample_momentum(state, rng)
    return state


def conditioned_diffusion_neg_log_dens_and_grad(obs_interval, num_steps_per_obs, y_seq, dim_u, dim_v_0, dim_v,
                                                forward_func, generate_x_0, generate_z, generate_σ, obs_func,
                                                use_gaussian_splitting=False, return_jax_funcs=False, *, device=0):
    """sde/mici_extensions.py:82-205: negative log target density and its gradient for the unconstrained-HMC comparator
    of the reference's experiments (to be used with a Mici `EuclideanMetricSystem`).  The model functions are the
    handles of one compiled model; `generate_σ` must be a number (fixed observation noise).  Returns
    `(neg_log_dens, grad_neg_log_dens)` with the reference's conventions: `grad_neg_log_dens(q) -> (grad, value)`, a
    non-finite value raises `HamiltonianDivergenceError`.  A leading batch axis on `q` evaluates that many points in
    one launch (the context is re-created when the batch size changes)."""
    from numbers import Number
    handles = (forward_func, generate_x_0, generate_z, obs_func)
    if not all(isinstance(h, ModelHandle) for h in handles) or len({h.model for h in handles}) != 1:
        raise TypeError("forward_func, generate_x_0, generate_z and obs_func must be the handles of one compiled model")
    if not isinstance(generate_σ, Number):
        raise NotImplementedError("variable observation noise (callable generate_σ) is not supported")
    if return_jax_funcs:
        raise NotImplementedError("there are no JAX functions behind this implementation")
    model = forward_func.model
    if (dim_u, dim_v_0, dim_v) != (model.dim_z, model.dim_v_0, model.dim_v):
        raise ValueError("dim_u, dim_v_0, dim_v do not match the compiled model")
    y = np.asarray(y_seq, dtype=np.float64).reshape(-1)
    ctxs = {}

    def _eval(q, want_grad):
        q = np.asarray(q, dtype=np.float64)
        q2 = np.atleast_2d(q)
        B = q2.shape[0]
        if B not in ctxs:
            for c in ctxs.values():
                c.close()
            ctxs.clear()
            ctxs[B] = ChmcContext(model.name, obs_interval, num_steps_per_obs, None, y, sigma=float(generate_σ),
                                  num_chains=B, device=device)
        val, g = ctxs[B].neg_log_dens_and_grad(q2, use_gaussian_splitting, want_grad)
        if not np.all(np.isfinite(val)):
            raise HamiltonianDivergenceError("Hamiltonian non-finite")
        if q.ndim == 1:
            return (float(val[0]), None if g is None else g[0])
        return val, g

    def neg_log_dens(q):
        return _eval(q, False)[0]

    def grad_neg_log_dens(q):
        val, g = _eval(q, True)
        return g, val

    return neg_log_dens, grad_neg_log_dens
