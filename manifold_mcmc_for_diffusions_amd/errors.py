"""Exception types of the reference's plugin surface (mici.errors); if Mici is installed its classes are reused
so that Mici's own transitions catch them."""
try:  # pragma: no cover - mici is not installed on the build / GPU boxes
    from mici.errors import (ConvergenceError, NonReversibleStepError, IntegratorError,
                             HamiltonianDivergenceError, AdaptationError)
except Exception:

    class Error(Exception):
        """Base class for errors."""

    class IntegratorError(Error, RuntimeError):
        """Error raised when integrator step fails."""

    class NonReversibleStepError(IntegratorError):
        """Error raised when integrator step fails reversibility check."""

    class ConvergenceError(IntegratorError):
        """Error raised when solver fails to converge within given number of iterations."""

    class AdaptationError(Error, RuntimeError):
        """Error raised when adaptation fails."""

    class HamiltonianDivergenceError(Error, RuntimeError):
        """Error raised when Hamiltonian diverges on a trajectory."""
