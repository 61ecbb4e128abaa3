"""`ConstrainedLeapfrogIntegrator` with Mici's constructor and `.step(state)` contract (mici 0.1.10, third-party to
the reference; call site scripts/utils.py:284-290), backed by the fused device step `chmc_leapfrog_step`.

Two execution paths, same results:
  * fused (default): when the projection solver is one of this package's two solvers the whole
    A(dt/2) B(dt) A(dt/2) step, with its `n_inner_step` inner h2-flow steps, runs on the device for all chains of the
    state at once;
  * composed: any other configuration (a custom solver callable or reverse-check norm) is composed on the host from
    the System methods exactly as Mici does, each of which calls the device library.
With one chain numerical failures raise `ConvergenceError` / `NonReversibleStepError`; with a batch they are
reported per chain in `state.step_status` (0 ok, 1 not converged, 2 diverged, 3 non-reversible) and a failed
chain keeps its previous position / momentum, as Mici's transitions discard a failed step.
"""
import numpy as np
from .errors import ConvergenceError, NonReversibleStepError, AdaptationError
from . import system as _system
from .system import (jitted_solve_projection_onto_manifold_newton,
                     jitted_solve_projection_onto_manifold_quasi_newton)


def maximum_norm(vct):
    """mici.solvers.maximum_norm"""
    return np.max(np.abs(vct))


class ConstrainedLeapfrogIntegrator:
    def __init__(self, system, step_size=None, n_inner_step=1, reverse_check_tol=2e-8,
                 reverse_check_norm=maximum_norm, projection_solver=jitted_solve_projection_onto_manifold_newton,
                 projection_solver_kwargs=None):
        self.system = system
        self.step_size = step_size
        self.n_inner_step = n_inner_step
        self.reverse_check_tol = reverse_check_tol
        self.reverse_check_norm = reverse_check_norm
        self.projection_solver = projection_solver
        self.projection_solver_kwargs = {} if projection_solver_kwargs is None else dict(projection_solver_kwargs)

    # ---- fused device path
    def _fusable(self):
        return (self.n_inner_step >= 1 and self.reverse_check_norm is maximum_norm and self.projection_solver in (
            jitted_solve_projection_onto_manifold_newton, jitted_solve_projection_onto_manifold_quasi_newton))

    def _step_fused(self, state):
        sysm, c = self.system, self.system.ctx
        single = np.ndim(state.pos) == 1
        sysm._sync(state, mom=True)
        kw = self.projection_solver_kwargs
        dt = np.asarray(state.dir, dtype=np.float64) * self.step_size
        res = c.leapfrog_step(
            dt, n_inner_step=self.n_inner_step, newton=self.projection_solver is jitted_solve_projection_onto_manifold_newton,
            constraint_tol=kw.get("constraint_tol", 1e-8), position_tol=kw.get("position_tol", 1e-8),
            divergence_tol=kw.get("divergence_tol", 1e10), max_iters=kw.get("max_iters", 50),
            reverse_check_tol=self.reverse_check_tol)
        q, p, _, _ = c.get_state(want_x_obs=False)
        state.pos = q[0] if single else q
        state.mom = p[0] if single else p
        sysm._adopt(state)
        state.step_status = res["status"]
        state.step_info = res
        i_f, i_b = int(res["iters_fwd"].sum()), int(res["iters_bwd"].sum())
        cc = state._call_counts
        newton = self.projection_solver is jitted_solve_projection_onto_manifold_newton
        for k in (("constr", "jacob_constr_blocks", "lu_jacob_product_blocks") if newton else ("constr",)):
            cc[k] = cc.get(k, 0) + i_f + i_b
        if single and res["status"][0] != 0:
            st = int(res["status"][0])
            if st == 3:
                raise NonReversibleStepError(
                    "Non-reversible step. Distance between initial and forward-backward integrated positions = "
                    f"{res['rev_err'][0]:.1e}.")
            name = "Newton" if newton else "Quasi-Newton"
            if st == 2:
                raise ConvergenceError(f"{name} iteration diverged.")
            raise ConvergenceError(f"{name} iteration did not converge.")
        return state

    # ---- composed path (Mici's own structure)
    def _h2_flow_retraction_onto_manifold(self, state, state_prev, dt):
        self.system.h2_flow(state, dt)
        self.projection_solver(state, state_prev, dt, self.system, **self.projection_solver_kwargs)

    def _project_onto_cotangent_space(self, state):
        state.mom = self.system.project_onto_cotangent_space(state.mom, state)

    def _step_a(self, state, dt):
        self.system.h1_flow(state, dt)
        self._project_onto_cotangent_space(state)

    def _step_b(self, state, dt):
        dt_i = dt / self.n_inner_step
        for i in range(self.n_inner_step):
            state_prev = state.copy()
            self._h2_flow_retraction_onto_manifold(state, state_prev, dt_i)
            if i == self.n_inner_step - 1:
                self.system.dh1_dpos(state)  # pre-evaluate: one evaluation fills J, factors, log-det and gradient
            self._project_onto_cotangent_space(state)
            state_back = state.copy()
            self._h2_flow_retraction_onto_manifold(state_back, state, -dt_i)
            rev_diff = self.reverse_check_norm(state_back.pos - state_prev.pos)
            if rev_diff > self.reverse_check_tol:
                raise NonReversibleStepError(
                    "Non-reversible step. Distance between initial and forward-backward integrated positions = "
                    f"{rev_diff:.1e}.")

    def _step(self, state, dt):
        self._step_a(state, 0.5 * dt)
        self._step_b(state, dt)
        self._step_a(state, 0.5 * dt)

    def step(self, state):
        """Integrator.step: copy the state, integrate by `state.dir * step_size`, return the copy."""
        if self.step_size is None:
            raise AdaptationError("Integrator `step_size` must be set before calling `step`.")
        state = state.copy()
        if self._fusable():
            return self._step_fused(state)
        self._step(state, state.dir * self.step_size)
        return state


__all__ = ["ConstrainedLeapfrogIntegrator", "maximum_norm", "ConvergenceError", "NonReversibleStepError",
           "jitted_solve_projection_onto_manifold_newton", "jitted_solve_projection_onto_manifold_quasi_newton",
           "_system"]
