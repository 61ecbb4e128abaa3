"""Synthetic FitzHugh-Nagumo workloads of BASELINE.json (host-side set-up shared by bench.py, the examples and the
multi-GPU driver): data simulation, contexts, initial states, momentum refresh, static-trajectory transitions."""
import numpy as np
from . import example_models as em
from .context import ChmcContext
from .init import fhn_initial_states, fhn_initial_states_device

SEED = 20200710  # scripts/utils.py:75-77
# daily counts of the boarding-school influenza outbreak (the reference's scripts/sir_model_boarding_school_data.npz,
# loaded at scripts/sir_model_chmc_experiment.py:62-64; obs_interval 1.0)
BOARDING_SCHOOL_COUNTS = (3, 8, 28, 75, 221, 281, 255, 235, 190, 125, 70, 28, 12, 5)


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

    def bytes_per_chain_step(self, k_iters, partition=None, newton=True):
        """Algorithmic bytes of one chain-step, SURVEY.md section 8(d): 8 [k (4 nnz + 8 Q) + 10 nnz + 25 Q]
        (quasi-Newton: k (nnz + 7 Q) instead of the Newton term)."""
        nnz = self.nnz(partition)
        it = (4 * nnz + 8 * self.ctx.Q) if newton else (nnz + 7 * self.ctx.Q)
        return 8.0 * (k_iters * it + 10 * nnz + 25 * self.ctx.Q)

    def nnz(self, partition=None):
        c = self.ctx
        p = c.partition if partition is None else partition
        return (c.C[p] * c.U + sum(b["nrows"] * b["ncols"] for b in c.blocks[p])
                + (c.T if c.noisy else 0))


class SirWorkload(FhnWorkload):
    """SIR chains on the boarding-school data as scripts/sir_model_chmc_experiment.py sets them up (BASELINE.json
    configs[3]): 14 daily counts, S steps per observation, ONE sub-sequence of R = 14 observations (dense 14 x 14 Gram
    block), sigma_y = 1, initial states by the Adam-based finder of the noisy system (sde/mici_extensions.py:1679-1801)
    with one generator for the whole batch (the finder restarts chains, so draws are not chain-indexed)."""

    def __init__(self, num_chains, num_steps_per_obs=200, sigma=1.0, device=0, chain_offset=0, total_chains=None,
                 use_gaussian_splitting=False, seed=SEED, adam_step_size=1e-1, log=None):
        from . import init
        self.B, self.S, self.T, self.R = num_chains, num_steps_per_obs, len(BOARDING_SCHOOL_COUNTS), len(BOARDING_SCHOOL_COUNTS)
        self.sigma, self.obs_interval = sigma, 1.0
        self.seed, self.chain_offset = seed, chain_offset
        self.y = np.asarray(BOARDING_SCHOOL_COUNTS, dtype=np.float64).reshape(-1, 1)
        self.ctx = ChmcContext("sir", self.obs_interval, num_steps_per_obs, self.R, self.y[:, 0], sigma=sigma,
                               use_gaussian_splitting=use_gaussian_splitting, num_chains=num_chains, device=device)
        rng = np.random.default_rng(np.random.SeedSequence(seed).spawn(chain_offset + 1)[chain_offset])
        _, _, self.init_tries = init.find_initial_states_by_gradient_descent_noisy_system(
            self.ctx, rng, adam_step_size=adam_step_size, max_iters=5000, log=log)
        self.rngs = [rng]
        self.solver = dict(newton=True, constraint_tol=1e-9, position_tol=1e-8, divergence_tol=1e10, max_iters=50,
                           reverse_check_tol=2e-8)
        self._torch_gen = None
