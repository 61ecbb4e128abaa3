"""Host-side (NumPy) description of the example diffusions, mirroring the attribute surface of
`sde.example_models.fhn` / `sde.example_models.sir` in the reference.

The attributes `forward_func`, `generate_x_0`, `generate_z`, `obs_func` are *model handles*: they are callable
with NumPy arrays (used for data simulation, initial states and trace functions on the host) and they carry the
model id that selects the compiled HIP device functions when passed to
`ConditionedDiffusionConstrainedSystem` -- callables cannot cross the C ABI.

Closed forms: SURVEY.md Appendix B, re-derived symbolically by tools/gen_models.py (tests/test_models.py checks
the two against each other).  Reference: sde/example_models/fhn.py:10-65, sde/example_models/sir.py:9-93.
"""
import numpy as np


class ModelHandle:
    """A host callable tagged with the device model it corresponds to."""

    def __init__(self, model, role, fn):
        self.model, self.role, self._fn = model, role, fn
        self.__name__ = f"{model.name}.{role}"

    def __call__(self, *a, **k):
        return self._fn(*a, **k)


class _Fhn:
    name, model_id = "fhn", 0
    dim_x, dim_w, dim_z, dim_v_0, dim_v, dim_y = 2, 1, 4, 2, 2, 1

    @staticmethod
    def _forward(z, x, v, δ):
        σ, ε, γ, β = z[..., 0], z[..., 1], z[..., 2], z[..., 3]
        x0, x1 = x[..., 0], x[..., 1]
        a0 = (x0 - x0 ** 3 - x1) / ε
        a1 = γ * x0 - x1 + β
        δζ = δ ** 1.5 * (v[..., 0] + v[..., 1] / np.sqrt(3.0)) / 2
        return np.stack([
            x0 + δ * a0 + (δ ** 2 / 2) * ((1 - 3 * x0 ** 2) * a0 - a1) / ε - (σ / ε) * δζ,
            x1 + δ * a1 + σ * np.sqrt(δ) * v[..., 0] + (δ ** 2 / 2) * (γ * a0 - a1) - σ * δζ,
        ], -1)

    @staticmethod
    def noise_matrix(z, x, δ):
        """d forward_func / d v (independent of x and v for this model); shape [..., X, V]."""
        σ, ε = z[..., 0], z[..., 1]
        c = δ ** 1.5 / 2
        A = np.empty(np.broadcast(σ, x[..., 0]).shape + (2, 2))
        A[..., 0, 0] = -(σ / ε) * c
        A[..., 0, 1] = -(σ / ε) * c / np.sqrt(3.0)
        A[..., 1, 0] = σ * np.sqrt(δ) - σ * c
        A[..., 1, 1] = -σ * c / np.sqrt(3.0)
        return A

    @staticmethod
    def _generate_z(u):
        u = np.asarray(u)
        return np.stack([np.exp(u[..., 0]), np.exp(u[..., 1]), np.exp(u[..., 2]), u[..., 3]], -1)

    @staticmethod
    def _generate_x_0(z, v_0):
        out = np.array(v_0, dtype=np.float64, copy=True)
        out[..., 1] -= z[..., 3]
        return out

    @staticmethod
    def _obs(x_seq):
        return x_seq[..., 0:1]


class _FhnNb(_Fhn):
    """FitzHugh-Nagumo as set up in the reference's notebook (FitzHugh-Nagumo_example.ipynb cells 7-18): the same
    drift, diffusion and strong-order-1.5 step, with the notebook's priors."""
    name, model_id = "fhn_nb", 2

    @staticmethod
    def _generate_z(u):
        u = np.asarray(u)
        return np.stack([np.exp(0.5 * u[..., 0] - 1), np.exp(0.5 * u[..., 1] - 2), 0.5 * u[..., 2] + 1,
                         0.5 * u[..., 3] + 1], -1)

    @staticmethod
    def _generate_x_0(z, v_0):
        return np.asarray(v_0, dtype=np.float64) - 0.5


class _Sir:
    name, model_id = "sir", 1
    dim_x, dim_y, dim_w, dim_z, dim_v_0, dim_v = 3, 1, 3, 4, 1, 3
    N = 763.0

    @staticmethod
    def _coeffs(z, y):
        N = _Sir.N
        β, γ, ζ, ϵ = z[..., 0], z[..., 1], z[..., 2], z[..., 3]
        y0, y1, y2 = y[..., 0], y[..., 1], y[..., 2]
        α = np.exp(y2)
        a = np.stack([
            -(α / N) * (np.exp(y1) + 0.5 * np.exp(y1 - y0)),
            (α / N) * (np.exp(y0) - 0.5 * np.exp(y0 - y1)) - β - 0.5 * β * np.exp(-y1),
            γ * (ζ - y2)], -1)
        B = np.zeros(a.shape[:-1] + (3, 3))
        B[..., 0, 0] = np.exp((-y0 + y1 + y2) / 2) / np.sqrt(N)
        B[..., 1, 0] = -np.exp((y0 - y1 + y2) / 2) / np.sqrt(N)
        B[..., 1, 1] = np.sqrt(β) * np.exp(-y1 / 2)
        B[..., 2, 2] = ϵ
        return a, B

    @staticmethod
    def _forward(z, x, v, δ):
        xc = np.array(x, dtype=np.float64, copy=True)
        xc[..., :2] = np.maximum(xc[..., :2], -500.0)
        a, B = _Sir._coeffs(z, xc)
        xn = xc + δ * a + np.sqrt(δ) * np.einsum("...ij,...j->...i", B, v)
        for k in (0, 1):
            xn[..., k] = np.where(xc[..., k] > -500.0, xn[..., k], xc[..., k])
        return xn

    @staticmethod
    def noise_matrix(z, x, δ):
        _, B = _Sir._coeffs(z, x)
        return np.sqrt(δ) * B

    @staticmethod
    def _generate_z(u):
        u = np.asarray(u)
        return np.stack([np.exp(u[..., 0]), np.exp(u[..., 1]), u[..., 2],
                         np.exp(np.sqrt(0.75) * u[..., 3] + 0.5 * u[..., 1] - 3)], -1)

    @staticmethod
    def _generate_x_0(z, v_0):
        v_0 = np.asarray(v_0)
        out = np.empty(v_0.shape[:-1] + (3,))
        out[..., 0] = np.log(762.0)
        out[..., 1] = 0.0
        out[..., 2] = v_0[..., 0]
        return out

    @staticmethod
    def _obs(x_seq):
        return np.exp(x_seq[..., 1:2])


def _generate_sigma_y(u):
    """generate_σ_y (sde/example_models/fhn.py:46-47, sir.py:92-93): observation-noise scale exp(u[4]) of a parameter
    vector with dim_u = dim_z + 1 components."""
    return np.exp(np.asarray(u)[..., 4])


def _finish(cls):
    if cls.name != "fhn_nb":  # (the notebook's model has noiseless observations)
        cls.generate_σ_y = ModelHandle(cls, "generate_σ_y", _generate_sigma_y)
    cls.forward_func = ModelHandle(cls, "forward_func", cls._forward)
    cls.generate_z = ModelHandle(cls, "generate_z", cls._generate_z)
    cls.generate_x_0 = ModelHandle(cls, "generate_x_0", cls._generate_x_0)
    cls.obs_func = ModelHandle(cls, "obs_func", cls._obs)
    return cls


fhn = _finish(_Fhn)
fhn_nb = _finish(_FhnNb)
sir = _finish(_Sir)
MODELS = {"fhn": fhn, "sir": sir, "fhn_nb": fhn_nb}


def generate_x_seq(model, z, x_0, v_seq, δ):
    """fhn.generate_x_seq (sde/example_models/fhn.py:54-60): host scan, used to simulate data."""
    x = np.asarray(x_0, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    out = np.empty((len(v_seq), model.dim_x))
    for s in range(len(v_seq)):
        x = model._forward(z, x, v_seq[s], δ)
        out[s] = x
    return out


def simulate_fhn_observations(num_obs=100, obs_interval=0.2, num_steps_per_obs_data=10000, seed=20200710,
                              z_true=(0.3, 0.1, 1.5, 0.8), x_0_true=(-0.5, 0.2), sigma=None):
    """Synthetic FitzHugh-Nagumo data as generated by scripts/fhn_model_noiseless_obs_chmc_experiment.py:84-93
    (fine-grid simulation, observe the first component); optional additive observation noise drawn after `v`."""
    rng = np.random.default_rng(seed)
    δ = obs_interval / num_steps_per_obs_data
    v = rng.standard_normal((num_obs * num_steps_per_obs_data, 2))
    σ, ε, γ, β = (float(t) for t in z_true)
    x0, x1 = (float(t) for t in x_0_true)
    sq, d15, d2 = δ ** 0.5, δ ** 1.5, δ ** 2 / 2
    r3 = 3.0 ** 0.5
    y = np.empty((num_obs, 1))
    k = 0
    vl = v.tolist()
    for t in range(num_obs):  # plain Python floats: ~1e6 steps per second
        for _ in range(num_steps_per_obs_data):
            v0, v1 = vl[k]
            k += 1
            a0 = (x0 - x0 * x0 * x0 - x1) / ε
            a1 = γ * x0 - x1 + β
            dz = d15 * (v0 + v1 / r3) / 2
            x0, x1 = (x0 + δ * a0 + d2 * ((1 - 3 * x0 * x0) * a0 - a1) / ε - (σ / ε) * dz,
                      x1 + δ * a1 + σ * sq * v0 + d2 * (γ * a0 - a1) - σ * dz)
        y[t, 0] = x0
    if sigma is not None:
        y = y + sigma * rng.standard_normal(y.shape)
    return y
