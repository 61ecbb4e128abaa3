"""Synthetic FitzHugh-Nagumo workloads of BASELINE.json (host-side set-up shared by bench.py, the examples and the
multi-GPU driver): data simulation, contexts, initial states, momentum refresh, static-trajectory transitions."""
import numpy as np
from . import example_models as em
from .context import ChmcContext
from .init import fhn_initial_states, fhn_initial_states_device

SEED = 20200710  # scripts/utils.py:75-77


class FhnWorkload:
    """FHN chains on one device: B chains of the (T, S, R, sigma) configuration, set up the way
    scripts/fhn_model_noisy_obs_chmc_experiment.py does (simulated data, linear-interpolation initial states,
    Newton solver with the script tolerances)."""

    def __init__(self, num_chains, num_steps_per_obs=400, num_obs=100, num_obs_per_subseq=5, sigma=0.1,
                 obs_interval=0.2, device=0, chain_offset=0, total_chains=None, use_gaussian_splitting=False,
                 num_steps_per_obs_data=10000, seed=SEED, device_init=False):
        self.B, self.S, self.T, self.R = num_chains, num_steps_per_obs, num_obs, num_obs_per_subseq
        self.sigma, self.obs_interval = sigma, obs_interval
        self.seed, self.chain_offset = seed, chain_offset
        self.y = em.simulate_fhn_observations(num_obs, obs_interval, num_steps_per_obs_data, seed=seed, sigma=sigma)
        self.ctx = ChmcContext("fhn", obs_interval, num_steps_per_obs, num_obs_per_subseq, self.y[:, 0], sigma=sigma,
                               use_gaussian_splitting=use_gaussian_splitting, num_chains=num_chains, device=device)
        if device_init:  # same states, solved on the device (no [B, Q] host array, no 8 B Q byte upload)
            self.rngs = fhn_initial_states_device(self.ctx, em.fhn, self.y, seed=seed, chain_offset=chain_offset,
                                                  total_chains=total_chains)
        else:
            q, xo, self.rngs = fhn_initial_states(em.fhn, obs_interval, num_steps_per_obs, self.y, num_chains,
                                                  sigma is not None, seed=seed, chain_offset=chain_offset,
                                                  total_chains=total_chains)
            self.ctx.set_state(q, None, xo, 0)
        self.solver = dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10, max_iters=50,
                           reverse_check_tol=2e-8)  # scripts/utils.py:131-166
        self._torch_gen = None

    # momentum refresh: IndependentMomentumTransition -> system.sample_momentum (sde/mici_extensions.py:1256-1259)
    def refresh_momentum_host(self):
        p = np.stack([r.standard_normal(self.ctx.Q) for r in self.rngs])
        self.ctx.set_momentum(p)
        self.ctx.project_onto_cotangent_space()

    def refresh_momentum_device(self, torch, device):
        """N(0, I) drawn on the device (torch is only the RNG / allocator here), then projected by the library."""
        if self._torch_gen is None:
            self._torch_gen = torch.Generator(device=device)
            self._torch_gen.manual_seed(SEED + 17 * int(self.rngs[0].integers(1 << 30)))
            self._pbuf = torch.empty((self.B, self.ctx.Q), dtype=torch.float64, device=device)
        self._pbuf.normal_(generator=self._torch_gen)
        torch.cuda.synchronize(device)
        self.ctx.set_momentum_device(self._pbuf.data_ptr())
        self.ctx.project_onto_cotangent_space()

    def refresh_momentum(self):
        """Device-side refresh (Philox stream keyed by the workload seed; draw counter per call)."""
        self._draw = getattr(self, "_draw", 0) + 1
        self.ctx.sample_momentum(self.seed, self._draw, self.chain_offset)

    def step(self, dt, active=None):
        return self.ctx.leapfrog_step(dt, active=active, **self.solver)

    def bytes_per_chain_step(self, k_iters, partition=None):
        """Algorithmic bytes of one chain-step, SURVEY.md section 8(d): 8 [k (4 nnz + 8 Q) + 10 nnz + 25 Q]."""
        nnz = self.nnz(partition)
        return 8.0 * (k_iters * (4 * nnz + 8 * self.ctx.Q) + 10 * nnz + 25 * self.ctx.Q)

    def nnz(self, partition=None):
        c = self.ctx
        p = c.partition if partition is None else partition
        return (c.C[p] * c.U + sum(b["nrows"] * b["ncols"] for b in c.blocks[p])
                + (c.T if c.noisy else 0))
