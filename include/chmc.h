/* chmc.h -- C ABI of the MI355X-native batched constrained-HMC leapfrog library (libchmc_hip.so).
 *
 * The reference (thiery-lab/manifold-mcmc-for-diffusions) has no FFI of its own: its hot path is a set of
 * JAX-jitted closures behind Mici's duck-typed System / Integrator / projection-solver protocol.  Each entry
 * point below therefore replaces one of those closures / methods (file:line relative to the reference tree),
 * batched over `num_chains` independent chains.  All arrays are fp64, C-contiguous, chain-major; host
 * pointers unless the name ends in `_device`.  The caller owns its buffers for the duration of a call; the
 * context owns all device memory.  A context is bound to one device / one stream and is not re-entrant.
 *
 * Return value: 0 on success, <0 on API misuse or a HIP error (text via chmc_last_error()).  Numerical
 * outcomes are per-chain status codes, never a non-zero return.
 */
#ifndef CHMC_H
#define CHMC_H
#ifdef __cplusplus
extern "C" {
#endif

/* model ids: which `forward_func / generate_z / generate_x_0 / obs_func` set is compiled in
 * (sde/example_models/fhn.py, sde/example_models/sir.py) */
#define CHMC_MODEL_FHN 0
#define CHMC_MODEL_SIR 1
#define CHMC_MODEL_FHN_NOTEBOOK 2 /* FitzHugh-Nagumo with the priors of FitzHugh-Nagumo_example.ipynb (cells 7-18) */

/* per-chain status of a projection / leapfrog step (exception classes of the reference):
 *   0 ok
 *   1 ConvergenceError "did not converge"           sde/mici_extensions.py:1398-1402, 1472-1476
 *   2 ConvergenceError "diverged" (|c| > dtol, NaN) sde/mici_extensions.py:1393-1397, 1467-1471
 *   3 NonReversibleStepError                        mici ConstrainedLeapfrogIntegrator._step_b
 *  -1 chain was not active in the call */
#define CHMC_OK 0
#define CHMC_NOT_CONVERGED 1
#define CHMC_DIVERGED 2
#define CHMC_NON_REVERSIBLE 3
#define CHMC_INACTIVE (-1)

typedef struct chmc_ctx chmc_ctx;

/* Arguments of ConditionedDiffusionConstrainedSystem.__init__ (sde/mici_extensions.py:211-228) that can cross
 * a C ABI; the model callables are replaced by `model`.  The metric argument (identity, or block diagonal with a dense
 * dim_u x dim_u first block, :279-315) is set with chmc_set_metric. */
typedef struct chmc_config {
  int model;                  /* CHMC_MODEL_* */
  int num_obs;                /* T = y_seq.shape[0] (dim_y = 1) */
  int num_steps_per_obs;      /* S */
  int num_obs_per_subseq;     /* R; 0 or >= T: no partitioning (:321-324) */
  int noisy;                  /* generate_sigma (:353-358): 0 None (noiseless observations), 1 a number (`sigma`),
                               * 2 the model's generate_sigma_y(u) = exp(u[dim_z]) (fhn.py:46-47, sir.py:92-93):
                               * variable observation noise, dim_u = dim_z + 1, q = [u(dim_u) | v_0 | v_seq | n] */
  int use_gaussian_splitting; /* :273-278 */
  int num_chains;             /* B */
  int device;                 /* HIP device ordinal */
  double obs_interval;
  double sigma;               /* fixed observation-noise std (generate_sigma given as a Number, :354-358); noisy == 1 only */
  const double* y_seq;        /* [T] */
} chmc_config;

int chmc_create(const chmc_config* cfg, chmc_ctx** out);
void chmc_destroy(chmc_ctx* ctx);
const char* chmc_last_error(void);
const char* chmc_backend(void); /* "hip:gfx950" */

/* dims[16] = {B, Q, NV, U, X, T, S, num_partition, RM, Kmax, C(part 0), C(part 1), K(part 0), K(part 1), V, V0} */
int chmc_get_dims(const chmc_ctx* ctx, int* dims);
/* blocks of a partition (sde/mici_extensions.py:321-351): 12 ints per block
 * {obs0, nobs, first, last, row0, nrows, ny, col0, ncols, step0, nsteps, 0} */
int chmc_get_blocks(const chmc_ctx* ctx, int partition, int* out);

/* ---- chain state: ConditionedDiffusionHamiltonianState (sde/mici_extensions.py:1285-1320) ------------- */
/* Sets pos [B][Q], mom [B][Q] (may be NULL), x_obs_seq [B][T][X], partition, and evaluates everything the
 * reference caches on a state (jacob_constr_blocks, chol_gram_blocks, log_det_sqrt_gram, grad_log_det_sqrt_gram,
 * :1151-1184). */
int chmc_set_state(chmc_ctx* ctx, const double* q, const double* p, const double* x_obs_seq, int partition);
int chmc_get_state(chmc_ctx* ctx, double* q, double* p, double* x_obs_seq, int* partition);
/* find_initial_state_by_linear_interpolation (sde/mici_extensions.py:1479-1547) for all chains at once, on the
 * device: given u [B][U], v_0 [B][V0] and full states at the observation times x_obs_seq_init [B][T][X]
 * (generate_x_obs_seq_init of the scripts), solves per time step for the noise increments that make the discretised
 * path interpolate linearly between them (solve_for_v_seq :1503-1526), sets pos = [u | v_0 | v_seq | n = 0],
 * mom = 0, x_obs_seq = x_obs_seq_init and evaluates the state caches, like chmc_set_state. */
int chmc_init_linear_interpolation(chmc_ctx* ctx, const double* u, const double* v_0, const double* x_obs_seq_init,
                                   int partition);
/* system.metric = PositiveDefiniteBlockDiagonalMatrix((DensePositiveDefiniteMatrix(M_0), IdentityMatrix()))
 * (sde/mici_extensions.py:279-315): M_0 [U][U] symmetric positive definite acts on the u-part of the state, the rest
 * of the metric is the identity; NULL restores the identity metric.  Enters get_M_0_matrix (:794-798) in the Woodbury
 * cores, log_det_sqrt_metric_0 (:305-310, :809), delta_q = metric.inv @ delta_mu in both solvers (:1033-1041,
 * :1105-1113), h2 / dh2_dmom / h2_flow (:1202-1231), normal_space_component (:1243-1250) and sample_momentum
 * (:1256-1259).  Refreshes the cached factors of the current state when one has been set; the momentum is kept as it
 * is (project it again if it has to be tangent with respect to the new metric).  Errors: Gaussian splitting (the
 * reference raises ValueError :293-300), M_0 not symmetric positive definite. */
int chmc_set_metric(chmc_ctx* ctx, const double* M_0);
/* One leaf of a batched dynamic-integration (no-U-turn) tree (the caller of integrator.step in the reference:
 * mici.transitions.MultinomialDynamicIntegrationTransition, scripts/utils.py:292-301), fused into one pass over the state
 * the last chmc_leapfrog_step produced.  The tree vectors stay with the caller as device buffers (plain pointers:
 * sub_prop_q, sub_sum [B][Q]; ck_p, ck_sum, ck_end [D][B][Q]).  For every chain with run[c] != 0:
 *   sub_sum += mom;  sub_prop_q = pos if take[c];  ck_p[store_slot], ck_sum[store_slot] = mom, sub_sum (store_slot >= 0);
 * and for the n_check nested sub-tree spans that end at this leaf (span k starts at the leaf recorded in checkpoint
 * check_lo + k; slot check_lo holds the largest span, and the span of slot check_lo + k + 1 is the right half of span k):
 *   out[c][6k], out[c][6k + 1] = dh_dmom(ck_p[check_lo + k]) . rho, dh_dmom(mom) . rho with rho = sub_sum -
 *   ck_sum[check_lo + k] + ck_p[check_lo + k] (momentum sum of the span): the two sides of the no-U-turn criterion
 *   (dh_dmom = metric.inv @ mom :1204-1208);
 *   with ck_end != NULL, for k < n_check - 1 (spans of four leaves or more), Mici's additional sub-tree checks
 *   (do_extra_subtree_checks, the transition's default) across the two halves of the span -- first leaf a, last leaf m
 *   of the left half, first leaf m + 1 of the right half, this leaf b:
 *   out[c][6k + 2], out[c][6k + 3] = dh_dmom(p_a) . rho1, dh_dmom(p_{m+1}) . rho1, rho1 = momenta of a..m plus p_{m+1};
 *   out[c][6k + 4], out[c][6k + 5] = dh_dmom(p_m) . rho2, dh_dmom(p_b) . rho2,     rho2 = momenta of m+1..b plus p_m;
 *   afterwards ck_end[check_lo] = mom (this leaf ends the left half of the next larger span starting at slot check_lo).
 *   Entries that are not computed are 0.
 * out is [B][6 n_check] on the host (zeros for chains that did not run; may be NULL when n_check == 0); n_check <= 10.
 * Sums are formed in a fixed order (no atomics).  run / take are host arrays (take is decided by the caller from
 * chmc_hamiltonian of the same state: multinomial sampling of the sub-tree's proposal); chmc_tree_step is the variant
 * that takes these decisions on the device. */
int chmc_tree_leaf(chmc_ctx* ctx, const int* run, const int* take, void* sub_prop_q_dev, void* sub_sum_dev,
                   void* ck_p_dev, void* ck_sum_dev, void* ck_end_dev, int store_slot, int check_lo, int n_check,
                   double* out);
/* The per-chain decisions of the same transition taken on the device, so that a leaf of the batched trees costs one
 * call and one 4-byte read-back instead of three host round trips (mici _build_tree's base case and its termination
 * logic, scripts/utils.py:292-301).  The library keeps per chain: h0 (Hamiltonian at the tree's root), alive (the tree
 * can still grow), run (the chain takes part in the current sub-tree), the sub-tree's multinomial log weight, the
 * number of leaves, the sum of the leaves' acceptance probabilities min(1, exp(h0 - h)), and the integrator-error /
 * divergence flags.
 *   chmc_tree_begin    h0 = Hamiltonian of the current states (returned in h0[B] if not NULL), alive = isfinite(h0),
 *                      counters and flags cleared;
 *   chmc_tree_set_alive overrides alive[B] (the caller ends a chain's tree on its whole-tree criterion);
 *   chmc_tree_subtree  run = alive, sub-tree weight = 0 (log weight -inf);
 *   chmc_tree_step     for the chains with run != 0: one chmc_leapfrog_step (same arguments), then
 *                      integrator error (status != 0) -> failed, alive = run = 0; else delta_h = h - h0 > max_delta_h or
 *                      NaN -> diverged, alive = run = 0; else the leaf joins the sub-tree: n_step += 1, sum_acc +=
 *                      min(1, exp(h0 - h)), weight += exp(-h), take = u_leaf[c] < exp(-h) / weight; then the
 *                      bookkeeping of chmc_tree_leaf with these run / take (same buffers and slot arguments), and a
 *                      negative criterion value on any checked span sets alive = run = 0 (no-U-turn termination).
 *                      *n_running = number of chains with run != 0 afterwards;
 *   chmc_tree_get      per-chain state to the host (any pointer may be NULL). */
int chmc_tree_begin(chmc_ctx* ctx, double* h0);
int chmc_tree_set_alive(chmc_ctx* ctx, const int* alive);
int chmc_tree_subtree(chmc_ctx* ctx);
int chmc_tree_step(chmc_ctx* ctx, const double* dt, int n_inner_step, int newton, double constraint_tol, double position_tol,
                   double divergence_tol, int max_iters, double reverse_check_tol, const double* u_leaf, double max_delta_h,
                   void* sub_prop_q_dev, void* sub_sum_dev, void* ck_p_dev, void* ck_sum_dev, void* ck_end_dev,
                   int store_slot, int check_lo, int n_check, int* n_running);
int chmc_tree_get(chmc_ctx* ctx, int* alive, int* run, int* n_step, int* failed, int* diverged, double* sub_logw,
                  double* sum_acc);
/* The per-doubling logic of the same transition on the device (mici MultinomialDynamicIntegrationTransition.sample: the
 * loop over tree depths around _build_tree), so that a doubling costs two calls with an 8-byte read-back each instead of
 * per-chain host logic on [B][Q] vectors.  The tree vectors stay the caller's device buffers [B][Q].
 *   chmc_tree_doubling_begin  for every live chain: direction fwd = u_dir[c] < 0.5; if the context's chain state sits on
 *                             the other tree edge, the edge (neg or pos q, p) is copied into the state and its caches
 *                             are re-evaluated; a new sub-tree begins (chmc_tree_subtree), sub_sum = 0.
 *                             *n_alive = chains whose tree can still grow (0: the transition is over).
 *   (2^depth chmc_tree_step calls with dt = +-step_size by the same comparison)
 *   chmc_tree_doubling_end    for every chain whose sub-tree completed (run != 0): biased progressive sampling, the
 *                             sub-tree's proposal replaces the tree's if u_accept[c] < min(1, exp(sub_logw - logw));
 *                             logw = logaddexp(logw, sub_logw); sum_mom += sub_sum; the chain's current state becomes
 *                             the tree's edge on the side of the doubling; then the no-U-turn criterion on the whole tree,
 *                             dh_dmom(neg edge) . sum_mom < 0 or dh_dmom(pos edge) . sum_mom < 0 -> alive = 0 (sums in a
 *                             fixed order).  *n_alive as above.
 *   chmc_tree_get_doubling    per chain: whether the proposal moved, doublings completed, log weight of the tree. */
int chmc_tree_doubling_begin(chmc_ctx* ctx, const double* u_dir, const void* neg_q_dev, const void* neg_p_dev,
                             const void* pos_q_dev, const void* pos_p_dev, void* sub_sum_dev, int* n_alive);
int chmc_tree_doubling_end(chmc_ctx* ctx, const double* u_accept, int depth, void* prop_q_dev, const void* sub_prop_q_dev,
                           void* sum_mom_dev, const void* sub_sum_dev, void* neg_q_dev, void* neg_p_dev, void* pos_q_dev,
                           void* pos_p_dev, int* n_alive);
int chmc_tree_get_doubling(chmc_ctx* ctx, int* moved, int* depth, double* logw);
int chmc_set_momentum(chmc_ctx* ctx, const double* p);
int chmc_get_state_device(chmc_ctx* ctx, void* q_dev, void* p_dev);  /* device-to-device copies */
int chmc_set_momentum_device(chmc_ctx* ctx, const void* p_dev);
/* system.sample_momentum(state, rng) (:1256-1259) with a device-side counter-based generator: mom = P(q) n,
 * n ~ N(0, I) from Philox4x32-10 keyed by `seed`, counter (component pair, `draw`, chain_offset + chain): the
 * stream of a chain does not depend on how chains are sharded over GPUs. */
int chmc_sample_momentum(chmc_ctx* ctx, unsigned long long seed, unsigned long long draw, int chain_offset);
/* Caller support for an accept / reject step around a trajectory (the role of Mici's transitions, which keep the
 * start state and discard a rejected or failed trajectory): device-side copy of (pos, mom) and masked restore
 * (mask [B], non-zero = restore; the state caches of restored chains are re-evaluated).  chmc_get_head copies the
 * first n components of every chain's position ([B][n]: u and v_0 live there, cf. trace_func of the scripts). */
int chmc_snapshot(chmc_ctx* ctx);
int chmc_restore(chmc_ctx* ctx, const int* mask);
/* The same from caller-held device buffers q_dev, p_dev [B][Q] (e.g. the tree edges of a dynamic transition): chains
 * with mask != 0 get (pos, mom) from the buffers and their state caches re-evaluated; momentum_is_tangent tells the
 * library whether those momenta lie in the cotangent space of their positions (see chmc_leapfrog_step). */
int chmc_restore_device(chmc_ctx* ctx, const void* q_dev, const void* p_dev, const int* mask, int momentum_is_tangent);
int chmc_get_head(chmc_ctx* ctx, int n, double* out);
/* system.update_x_obs_seq(state) (:1240-1241, :384-397) */
int chmc_update_x_obs_seq(chmc_ctx* ctx);
/* SwitchPartitionTransition.sample (:1279-1282): partition = (partition + 1) % num_partition, update x_obs_seq,
 * re-evaluate the state caches */
int chmc_switch_partition(chmc_ctx* ctx);

/* ---- per-op entry points, evaluated at the current state ------------------------------------------------
 * Internally the library keeps the dc/dv rows of a state in a compact factored form (per step a row-independent X x V
 * matrix, per observation interval the RM x X adjoint frame; DESIGN.md section 4 "Compact rows") and the stepping path
 * never writes the full rows of blocks with at most 8 rows.  The entry points below that return rows or multiply by them
 * rebuild the row-slot array first (one extra pass, only when called); results are the same to rounding.
 *
 * Environment switches -- EVERY variable the library reads; each is exercised by a GPU test.  Defaults are chosen from the
 * layout (blocks per chain, block length, rows) alone, never from the number of chains -- the one exception, the execution
 * model of single-block layouts (CHMC_RETRACT_KERNEL), chooses between bit-identical paths --, so a chain's results do not depend
 * on the shard it runs in (bitwise: tests/test_hip_parity.py::test_results_do_not_depend_on_the_shard_size).  Pinning a
 * switch to a non-default value changes bits at the rounding level (summation order; the time-parallel scan equals the
 * sequential recursion to about 1e-15 relative after its final sweep, not bitwise), never statuses.
 *   read by chmc_create:
 *   CHMC_COMPACT_ROWS=0     round 1's stored-rows kernel family everywhere (the A/B partner of the default)
 *   CHMC_GRAM_MFMA=1        fp64-MFMA Gram kernel for 16-row blocks (on the stored-rows Newton sweep)
 *   CHMC_PAR_SCAN=0/1       time-parallel forward scan off / forced (default: at most 4 blocks per chain, >= 1024 steps)
 *   CHMC_PAR_WAVES=1/2/4    wavefronts per (chain, block) of that scan (default: from the block length; also read by the
 *                           comparator target's scan at every call)
 *   CHMC_ROW_SPLIT=1/2/4    16-row state evaluation: 1 = stored-rows sweeps, otherwise interval-parallel (default: <= 4 blocks)
 *   CHMC_HALVES=2           two overlapped half-batches per step
 *   read at every call:
 *   CHMC_NO_FWD_SCAN=1      generic functor instead of the hand-scheduled forward scan
 *   CHMC_STEP_FUSIONS=0     the momentum correction and the reverse flow of a step as passes of their own instead of inside
 *                           the J p / J^T lambda passes (same bits)
 *   CHMC_RETRACT_KERNEL=0/1/2  one 16-row block per chain: batched launches / one workgroup of 8 wavefronts per chain / of 4
 *                           wavefronts (two chains per compute unit).  Default: 8 up to one chain per compute unit, 4 up to
 *                           four, batched beyond; all three give the same bits */
int chmc_constr(chmc_ctx* ctx, double* c);                                  /* :473-519, :1151-1155  [B][C] */
/* :521-624, :1157-1161.  dc_du [B][C][U]; dc_dv [B][RM][NV] row-slot layout (slot i = row i of the block that
 * owns the column); dc/dn is sigma on observation rows (:601-608). */
int chmc_jacob_constr_blocks(chmc_ctx* ctx, double* dc_du, double* dc_dv);
int chmc_chol_gram_blocks(chmc_ctx* ctx, double* chol_C, double* chol_D);   /* :626-687  [B][U][U], [B][K][RM][RM] */
int chmc_log_det_sqrt_gram(chmc_ctx* ctx, double* out);                     /* :800-820, :1169-1171  [B] */
int chmc_grad_log_det_sqrt_gram(chmc_ctx* ctx, double* grad);               /* :1143-1146, :1173-1184  [B][Q] */
int chmc_lmult_by_jacob_constr(chmc_ctx* ctx, const double* vct, double* out);   /* :822-877  [B][Q] -> [B][C] */
int chmc_rmult_by_jacob_constr(chmc_ctx* ctx, const double* vct, double* out);   /* :879-913  [B][C] -> [B][Q] */
int chmc_lmult_by_inv_gram(chmc_ctx* ctx, const double* vct, double* out);       /* :915-942  [B][C] -> [B][C] */
int chmc_normal_space_component(chmc_ctx* ctx, const double* vct, double* out);  /* :983-993, :1243-1250 */
int chmc_project_onto_cotangent_space(chmc_ctx* ctx);  /* :1252-1254, in place on the state's momentum */
int chmc_hamiltonian(chmc_ctx* ctx, double* h);        /* :1186-1202  [B][3] = {h1 + h2, q.q/2, p.p/2} */

/* conditioned_diffusion_neg_log_dens_and_grad (sde/mici_extensions.py:82-205), the target of the reference's
 * unconstrained-HMC comparator, for B independent points: q [B][QH], QH = U + V0 + T S V (no observation-noise part),
 *   value [B] = 1/2 sum_t ((y_t - obs_func(x_t)) / sigma)^2 + T log sigma (+ 1/2 q.q unless use_gaussian_splitting),
 *   grad [B][QH] (may be NULL).  Needs a context with observation noise; does not touch the chain states. */
int chmc_neg_log_dens_and_grad(chmc_ctx* ctx, const double* q, int use_gaussian_splitting, double* value, double* grad);
/* The same with q_dev [B][U + V0 + T S V] and grad_dev (may be NULL) in device memory; `value` [B] on the host. */
int chmc_neg_log_dens_and_grad_device(chmc_ctx* ctx, const void* q_dev, int use_gaussian_splitting, double* value,
                                      void* grad_dev);
/* The device-resident loop of find_initial_state_by_gradient_descent_noisy_system (sde/mici_extensions.py:1679-1801; its
 * objective :1706-1737 is the comparator's target for a fixed sigma): one Adam iteration is these two calls.
 * chmc_adam_objective_device: objective and gradient at u_v_dev [B][U + V0 + T S V] (gradient into grad_dev) and, in one
 *   read-back, out3 [B][3] = objective, |u_v|^2, 1 if every gradient entry is finite (else 0): what the restart rules
 *   (:1745-1765) need of the chain.
 * chmc_adam_update_device: the Adam step (jax.example_libraries.optimizers.adam, :1740) in place on u_v_dev, m_dev, v_dev:
 *   moments of every chain (non-finite gradient entries count as 0), parameters of chain c moved by
 *   coef[c][1] * m / (sqrt(v * coef[c][0]) + eps), coef [B][2] on the host = {1 / (1 - b2^t), lr / (1 - b1^t) or 0}. */
int chmc_adam_objective_device(chmc_ctx* ctx, const void* u_v_dev, void* grad_dev, double* out3);
int chmc_adam_update_device(chmc_ctx* ctx, void* u_v_dev, void* m_dev, void* v_dev, const void* grad_dev,
                            const double* coef, double b1, double b2, double eps);

/* Projection solvers (newton != 0: newton_projection :1065-1135 with its host wrapper :1405-1476;
 * newton == 0: quasi_newton_projection :999-1063 / :1323-1402).  Projects the points q [B][Q] onto the manifold
 * along the rows of the constraint Jacobian at the CURRENT state (= `state_prev` of the reference).
 * Outputs: q_out [B][Q]; mu_out [B][Q] = mu/dt (mu/sin dt with Gaussian splitting); iters, norm_dq, err, status [B].
 * The state itself is not modified. */
int chmc_project(chmc_ctx* ctx, int newton, const double* q, const double* dt, double constraint_tol,
                 double position_tol, double divergence_tol, int max_iters, double* q_out, double* mu_out,
                 int* iters, double* norm_dq, double* err, int* status);

/* One ConstrainedLeapfrogIntegrator.step per chain (mici 0.1.10): A(dt/2) B(dt) A(dt/2); B = n_inner_step inner steps
 * of dt / n_inner_step (scripts/utils.py:131-136 --num-inner-h2-step), each with forward projection, momentum
 * projection at the new point, reverse projection and reversibility check; dh1_dpos is evaluated at the last new point
 * only.  dt [B] = state.dir * step_size; active [B] (NULL: all).  A chain whose step fails -- in whichever inner step --
 * keeps its state (Mici discards the partial step).  iters_fwd / iters_bwd are summed over the inner steps, rev_err is
 * that of the last inner step a chain ran.
 * The two half-kicks A(dt/2): p <- P(q)[p - dt/2 dh1_dpos(q)] use the projected kick direction P(q) dh1_dpos(q) that
 * the library keeps with every evaluated state: for a momentum already in the cotangent space (after
 * chmc_sample_momentum, chmc_project_onto_cotangent_space or a previous step) P(q)[p - h g] = p - h P(q) g, which
 * saves two of the three passes over the stored Jacobian rows; a momentum set through chmc_set_state /
 * chmc_set_momentum(_device), or carried across chmc_switch_partition, takes the full projection as in the reference. */
int chmc_leapfrog_step(chmc_ctx* ctx, const double* dt, const int* active, int n_inner_step, int newton,
                       double constraint_tol, double position_tol, double divergence_tol, int max_iters,
                       double reverse_check_tol, int* status, int* iters_fwd, int* iters_bwd, double* rev_err);

/* A whole integration trajectory per chain: up to n_steps[c] (n_steps == NULL: n_steps_all for every chain) consecutive
 * ConstrainedLeapfrogIntegrator.step calls of the chains with active[c] != 0, the loop a Mici integration transition
 * runs around integrator.step (mici.transitions: `for s in range(n_step): state = integrator.step(state)`, ended by
 * an IntegratorError; scripts/utils.py:284-301).  Semantics per chain = chmc_leapfrog_step applied repeatedly: a chain
 * whose step fails keeps the state its last successful step produced, reports that step's status and takes no further
 * steps.  Outputs [B]: n_done (successful steps), status (0: all steps done; 1 / 2 / 3: the failing step's code; -1:
 * inactive), iters_fwd / iters_bwd (Newton iterations summed over the chain's steps, the failing one included), rev_err
 * (reverse-check distance of the last step that reached the check).
 * Implemented as one batched chmc_leapfrog_step per step over the chains still running (tests/test_trajectories.py: bitwise
 * the host loop).  For layouts with one 16-row block per chain every retraction inside those steps already runs per chain,
 * in the chain's own workgroup, for as many Newton iterations as THAT chain needs (k_retract_chain; lax.while_loop
 * :1119-1131).  (Round 3's asynchronous per-chain-phase engine behind this entry point lost to the batched steps at every
 * BASELINE shape and was removed in round 4.) */
int chmc_leapfrog_steps(chmc_ctx* ctx, const double* dt, const int* active, const int* n_steps, int n_steps_all,
                        int n_inner_step, int newton, double constraint_tol, double position_tol, double divergence_tol,
                        int max_iters, double reverse_check_tol, int* n_done, int* status, int* iters_fwd, int* iters_bwd,
                        double* rev_err);

/* The single collective of a chain-sharded run: chains are independent, every rank (one process per
 * GPU) steps its own contiguous shard with no communication, and a sampling segment ends with ONE all-gather of the
 * traced per-chain samples over RCCL / xGMI.  (The reference runs its chains sequentially in one process,
 * scripts/utils.py:351-363; there is no reference collective to mirror.)
 *   chmc_comm_unique_id : rank 0 creates the 128-byte RCCL id and hands it to the other ranks by whatever means the
 *                         launcher has (file, MPI, torch.distributed.broadcast_object_list ...)
 *   chmc_comm_init      : every rank, same id; creates the context's communicator (collective call)
 *   chmc_gather_samples : local_dev [count] doubles of this rank -> gathered_dev [world][count] on EVERY rank, rank-major
 *                         (equal shards); both are device buffers of the caller; enqueued on the context's stream and
 *                         complete when the call returns
 *   chmc_comm_info      : world size and rank READ BACK from the communicator (ncclCommCount / ncclCommUserRank): what a
 *                         launcher prints to show that the ranks it started really joined one RCCL communicator
 *   chmc_comm_destroy   : releases the communicator (chmc_destroy does the same for a context that still owns one).
 * N > 1 ranks have been rehearsed with gloo on CPU only so far (tests/test_distributed_gloo.py); see INTEGRATION.md. */
int chmc_comm_unique_id(void* id128);
int chmc_comm_init(chmc_ctx* ctx, const void* id128, int rank, int world);
int chmc_gather_samples(chmc_ctx* ctx, const void* local_dev, long count, void* gathered_dev);
int chmc_comm_info(chmc_ctx* ctx, int* world, int* rank);
int chmc_comm_destroy(chmc_ctx* ctx);

/* evaluation counters since creation: {constr, jacob_constr_blocks, lu_jacob_product_blocks, chol_gram_blocks,
 * grad_log_det_sqrt_gram, leapfrog_step calls, newton rounds launched (a launch of the per-chain retraction kernel counts
 * as one), 0 (reserved)} (cf. _call_counts, :1451-1461).
 * Host-side counts: the call does not touch the device. */
int chmc_get_counters(const chmc_ctx* ctx, long long* out8);

/* diagnostics of the kernel paths taken since creation (tests assert with them that an optional kernel family ran):
 *   out80[0]       blocks the time-parallel forward scan handed to its sequential fallback
 *   out80[1 .. 63] histogram of sweeps to convergence of the time-parallel scan (1 + sweeps + 16 (guess kind - 1)), [15] parked
 *   out80[64]      launches of the fp64-MFMA Gram kernel (v_mfma_f64_16x16x4_f64; 16-row blocks, CHMC_GRAM_MFMA=1)
 *   out80[65]      launches of the vector-FMA Gram kernel over stored rows (16-row blocks)
 *   out80[66]      launches of the per-chain retraction kernel (k_retract_chain: one 16-row block per chain)
 *   out80[67]      launches of the per-chain trajectory kernel (k_traj_chain: whole leapfrog steps of such a chain)
 *   out80[68 .. 79] reserved (0)
 * Synchronises the context's stream. */
int chmc_get_diagnostics(chmc_ctx* ctx, long long* out80);

/* ---- measurement: HIP events recorded on the library's own stream around every kernel launch ------------- */
/* kernel classes: 0 other, 1 newton_blk (constr + Jacobian + Gram/LU of a Newton iteration), 2 state_blk
 * (Jacobian store + Gram + Cholesky), 3 grad_log_det_blk, 4 update (J^T lambda column pass), 5 solve_chain,
 * 6 jacob_vec (J w), 7 constr (forward scan only), 8 element-wise, 9 sym_blk */
#define CHMC_NUM_KERNEL_CLASSES 10
int chmc_profile_enable(int on);  /* 0 stop, 1 all classes, else bit mask (1 << class); resets the accumulators */
int chmc_profile_stride(int every); /* time every `every`-th launch of a profiled class only (default 1); an event pair costs ~30 us of stream bubbles */
int chmc_profile_get(double* ms, long long* launches);  /* [CHMC_NUM_KERNEL_CLASSES] each; synchronises */

#ifdef __cplusplus
}
#endif
#endif
