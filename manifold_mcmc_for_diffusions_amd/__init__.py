"""MI355X-native batched constrained-HMC leapfrog step for partially observed diffusions.

Drop-in for the hot path of thiery-lab/manifold-mcmc-for-diffusions (`sde.mici_extensions` constrained system +
Mici's ConstrainedLeapfrogIntegrator): Python host keeping the Mici System / Integrator surface, thin ctypes C ABI
(include/chmc.h), hand-written HIP kernels for gfx950.  Importing the package never touches the GPU; creating a
system / context requires the built HIP library and a visible MI355X (there is no CPU fallback).
"""
from . import example_models  # noqa: F401
from .errors import (ConvergenceError, NonReversibleStepError, IntegratorError,  # noqa: F401
                     HamiltonianDivergenceError, AdaptationError)
from .context import ChmcContext  # noqa: F401
from .system import (  # noqa: F401
    ConditionedDiffusionConstrainedSystem, ConditionedDiffusionHamiltonianState, SwitchPartitionTransition,
    IdentityMatrix, DensePositiveDefiniteMatrix, PositiveDefiniteBlockDiagonalMatrix,
    jitted_solve_projection_onto_manifold_newton,
    jitted_solve_projection_onto_manifold_quasi_newton, find_initial_state_by_linear_interpolation,
    conditioned_diffusion_neg_log_dens_and_grad)
from .integrators import ConstrainedLeapfrogIntegrator  # noqa: F401
from .adapters import OnlineBlockDiagonalMetricAdapter  # noqa: F401

__version__ = "0.1.0"
