"""TEST INFRASTRUCTURE ONLY (see oracle/README): torch.func fp64 restatement of
`conditioned_diffusion_neg_log_dens_and_grad` (sde/mici_extensions.py:82-205), the target density of the reference's
unconstrained-HMC comparator, with autograd standing in for jax.value_and_grad.  Fixed observation noise sigma."""
import torch
from .models import MODELS


def neg_log_dens_and_grad(model, obs_interval, num_steps_per_obs, y_seq, sigma, q, use_gaussian_splitting=False):
    m = MODELS[model]
    y = torch.as_tensor(y_seq, dtype=torch.float64).reshape(-1, 1)
    T, S = y.shape[0], num_steps_per_obs
    dl = obs_interval / S
    q = torch.as_tensor(q, dtype=torch.float64).clone().requires_grad_(True)
    var_sigma = isinstance(sigma, str)  # "variable": sigma = generate_sigma(u) = exp(u[dim_z]), dim_u = dim_z + 1 (:163-164)
    U, V0, V = m.dim_z + int(var_sigma), m.dim_v_0, m.dim_v
    u, v_0, v_seq = q[:U], q[U:U + V0], q[U + V0:].reshape(T * S, V)
    if var_sigma:
        sigma = torch.exp(u[m.dim_z])
    z = m.generate_z(u)
    x = m.generate_x_0(z, v_0)
    obs = []
    for s in range(T * S):  # lax.scan of :180-186
        x = m.forward_func(z, x, v_seq[s], dl)
        if (s + 1) % S == 0:
            obs.append(m.obs_func(x))
    y_mean = torch.stack(obs).reshape(-1, 1)
    log_sigma = torch.log(sigma) if var_sigma else torch.log(torch.tensor(float(sigma), dtype=torch.float64))
    val = 0.5 * (((y - y_mean) / sigma) ** 2).sum() + T * log_sigma
    if not use_gaussian_splitting:
        val = val + 0.5 * (q ** 2).sum()
    (g,) = torch.autograd.grad(val, q)
    return float(val.detach()), g.detach().numpy()
