// Device-side work functions of the batched constrained-HMC leapfrog step.
//
// Every kernel is a functor with `operator()(int tid)`; the HIP backend (chmc.hip) launches
// them over a 1-D grid, one work item per lane.  Work items are (chain), (chain, block),
// (chain, block, row) or (chain, column) -- "block" being one conditionally independent
// sub-sequence of the observation sequence (sde/mici_extensions.py:321-351, 413-471).
//
// Data layout in HBM (B chains, all fp64, chain-major so a chain's state is contiguous):
//   q, p, grad   [B][Q]            q = [u(U) | v_0(V0) | v_seq(T*S*V) | n(T)]  (:476-484)
//   traj         [B][T*S+16 Kmax][X]  per block nsteps+1 states (x_s before step s), rows line-aligned (CHMC_TPAD)
//   Jv           [B][RM][NV]       "row-slot" Jacobian: slot i holds row i of the block that
//                                  owns the column, so J^T lambda / J w / Gram builds are
//                                  unit-stride streams over the v-part of q
//   JuP, E       [B][Kmax][RM][U]  dc/du rows and D^-1 dc/du, padded to RM rows per block
//   facD         [B][Kmax][RM][RM] Cholesky factor of D_b, identity-padded
//   facC, Cinv   [B][U][U]
// Two state slots (current state / proposal) are selected per chain through `cur[c]`, so
// accepting a step is a flag flip, not a copy.
//
// Reference lines cited as :NNN are sde/mici_extensions.py unless stated otherwise.
#pragma once
#include <math.h>
#include <stdint.h>
#include "chmc_model.h"

// functors that the per-chain kernels (chmc_retract.h) call from inside their phase loops: a real call there makes the kernel
// obey the function ABI, which cost the scan's sweep loop 180 spilled registers
#define CHMC_FI __attribute__((always_inline))
#if defined(__HIPCC__)
#define CHMC_UNROLL _Pragma("unroll")
#define CHMC_UNROLL4 _Pragma("unroll 4")  // (run-time trip counts: four iterations' loads in flight together)
#else
#define CHMC_UNROLL
#define CHMC_UNROLL4
#endif

namespace chmc {

#ifndef CHMC_FWD_DEPTH
#define CHMC_FWD_DEPTH 4  // tiles of noise increments in flight in the forward scans
#endif
#define CHMC_Q_PAD 256 // doubles of slack after every [B][Q] position buffer (see fwd_block)
// Trajectory layout [B][T S + CHMC_TPAD Kmax][X]: block b of a chain stores its nsteps + 1 states from entry step0 +
// CHMC_TPAD b on.  The padding keeps every block's row on a 128-byte line whenever its first step is (16 states are 256 /
// 384 bytes for X = 2 / 3): with the rows one state apart (round 1-3) seven of eight blocks started 16 bytes into a line,
// every 128-byte tile the forward scan's helper wavefront stores straddled two lines, and the counters showed 165 MB
// written for a 129 MB trajectory.
#define CHMC_TPAD 16

struct BlockDesc {
  int obs0, nobs, first, last;
  int row0, nrows, ny;
  int col0, ncols;
  int step0, nsteps;
  int pad_;
};

struct Sys {
  int B, T, S, noisy, gaussian;
  int varsig;  // observation noise sigma = generate_sigma(u) = exp(u[Z]) (fhn.py:46-47, sir.py:92-93): U = Z + 1
  int U, X, V, Z, V0;
  int Q, NV, K, C, Kmax, RM, TRJ, NCOL;  // NCOL = NV + (noisy ? T : 0)
  int NOBS;  // most observations in one block, over both partitions (interval frames LF)
  double dl, sigma;
  const double* y;
  const double* xobs;  // [B][T][X]
  const BlockDesc* blk;
  const int* obs2blk;
  const int* order;  // [B * K] work item -> chain * K + block of the wave-per-block kernels (longest blocks first)
  // metric M = diag(M_0, I) (:303-315): [3][U][U] = M_0, M_0^-1, lower Cholesky factor of M_0; nullptr = identity
  const double* m0;
  double hld_m0;  // log det(M_0) / 2 (log_det_sqrt_metric_0 :305-310)
};
// (M_0^-1 v_u)[a] and (M_0 v_u)[a] for the u-part of a [Q] vector (the rest of the metric is the identity)
CHMC_HD inline double metric_inv_u(const Sys& sy, const double* vu, int a) {
  if (!sy.m0) return vu[a];
  const double* W = sy.m0 + sy.U * sy.U + a * sy.U;
  double t = 0.0;
  for (int b = 0; b < sy.U; ++b) t += W[b] * vu[b];
  return t;
}
// generate_sigma(u) of a chain with position q: a number (:354-358) or exp(u[dim_z]) with variable observation noise
CHMC_HD inline double sigma_at(const Sys& sy, const double* q) { return sy.varsig ? exp(q[sy.Z]) : sy.sigma; }
CHMC_HD inline double metric_mul_u(const Sys& sy, const double* vu, int a) {
  if (!sy.m0) return vu[a];
  const double* Mr = sy.m0 + a * sy.U;
  double t = 0.0;
  for (int b = 0; b < sy.U; ++b) t += Mr[b] * vu[b];
  return t;
}

struct Slots {
  double* q[2];
  double* p[2];
  double* traj[2];
  double* JuP[2];
  double* Jv[2];
  double* facD[2];
  double* E[2];
  double* facC[2];
  double* Cinv[2];
  double* ldb[2];
  double* logdet[2];
  double* grad[2];
  double* pg[2];  // [B][Q] P(q) dh1_dpos(q): the projected gradient of the slot's state (see chmc_leapfrog_step)
  // Compact form of the stored dc/dv rows (blocks of at most 8 rows, wave kernels): inside observation interval m of a
  // block every row is the interval's frame applied to ONE row-independent matrix per step,
  //     dc_i/dv_s = LF[m][i] . PB[s],   PB[s] = Pf E_s B_s  (X x V),   LF[m][i] = the adjoint row i at the interval's end,
  // so the passes of a Newton iteration over the previous point's Jacobian (J^T lambda in KUpdatePB, the Gram
  // contraction in k_newton_lean) read X V doubles per step instead of up to RM V.  Null when not in use.
  double* PB[2];  // [B][T S][X V]
  double* LF[2];  // [B][Kmax][NOBS][RM][X]
  int* cur;
};

struct Work {
  double* muF;      // [B][Kmax][NOBS][X]  sum_i lambda_i LF[m][i]: the multipliers applied to the interval frames
  double* muF2;     //                     the same for lampad2 (two-vector projection)
  double* ivl;      // [B][Kmax][NOBS][2 X X + X Z]  per-interval sums of the two-phase Newton sweep (k_newton_ivl)
  double* trajw;    // [B][TRJ]      trajectory of the Newton iterate
  double* cpad;     // [B][Kmax][RM] constraint values, block-padded
  double* cpad2;    // [B][Kmax][RM] second right-hand side of a two-vector projection
  double* lampad2;  // [B][Kmax][RM] multipliers of the first vector of a two-vector projection
  double* tpad;     // [B][Kmax][RM] D_b^-1 (rhs)_b
  double* lampad;   // [B][Kmax][RM] multipliers
  double* Ew;       // [B][Kmax][RM][U] D_b^-1 dc/du (current iterate)
  double* Cb;       // [B][Kmax][U][U]
  double* sb;       // [B][Kmax][U]
  double* mu;       // [B][Q]        staging for the multiplier output of chmc_project
  double* qb;       // [B][Q]        reverse-check iterate
  double* pb;       // [B][Q]
  double* vin;      // [B][Q]        generic input vector (per-op API)
  double* Xd;       // [B][T*S][RM][X] tangents for grad log det
  double* gup;      // [B][Kmax][U]
  double* JvW;      // [B][RM][NV]   rows of the current Newton iterate (16-row blocks only: Gram formed by k_gram_rows)
  double* Dw;       // [B][Kmax][RM][RM] Gram block of the current evaluation (wave kernels -> factor kernels)
  double* JuL;      // [B][Kmax][RM][U]  dc/du rows of the current Newton iterate
  double* zbP;      // [B][Kmax][RM][Z]  dc/dz rows (before the generate_z chain rule) of the last state evaluation
  double* gMb;      // [B][Kmax][RM][RM] (G^-1)_bb
  double* gzd;      // [B][Kmax][RM][Z]  generate_z'(u) (G^-1 dc/du)_i
  double* gWu;      // [B][Kmax][RM][U]  rows of G^-1 dc/du
  double* gxdt;     // [B][Kmax][RM][X]  tangents at the rows' terminal times
  double* sdt;      // [B] sin(dt)
  double* cdt;      // [B] cos(dt)
  double* err;      // [B]
  unsigned long long* ndq;  // [B] bit pattern of max |dq|
  unsigned long long* rev;  // [B] bit pattern of reverse-check distance
  double* dt;       // [B]
  double* part;     // [B][NPART] partial sums
  int* iters;       // [B]
  int* nw;          // [B] Newton loop: 0 finished, 1 iterating, 2 the time-parallel forward scan of the current iterate has
                    //     not settled yet: the chain sits this round out and its scan goes on in the next (k_fwd_par, K = 1)
                    //     forward-retraction mask in `nw`; one merged scan serves both groups, each chain with its own
                    //     (slot, iterate) selection.  Null outside the engine.
  int* ok;          // [B] chain still good in this step
  int* status;      // [B]
  int* nstat;       // [B] status of last projection
  int* n_active;    // [4] chains still in the Newton loop after round r, in slot r & 3 (KCheck of round r clears the next
                    //     slot: no memset between the rounds); per view: [batch | half 0 | half 1] x 4
  double* gcq;      // [B][Kmax][NOBS][2 X X + X Z] interval-parallel grad-log-det: C1 | C2 | Q0 of every interval (few long blocks)
  double* gbw;      // [B][Kmax][NOBS][X + 2 Z]     its backward sweep: x-bar handed on | z-bar sums of the two phases
  unsigned* ticket; // [B] workgroups of a column-max launch that have finished a chain (the last one runs the launch's
                    //     per-chain epilogue: KUpdatePB's fused convergence check)
  const double* zeros;  // [256] zeros (stand-in source for loads of structurally zero Jacobian entries)
  int* nfallback;   // [1] blocks the time-parallel forward scan handed to its sequential fallback (diagnostic)
};

// slot selection as an explicit select: indexing the by-value kernel-argument pointer pairs with a run-time slot
// makes the compiler spill the whole argument block to scratch
template <class T_>
CHMC_HD inline T_* pick(T_* const (&a)[2], int s) {
  return s ? a[1] : a[0];
}

// Newton-phase kernels: is chain c part of the launch?  (prev / qsel are the launch's own arguments; the by-reference
// parameters remain from round 3's asynchronous engine, whose merged launches picked them per chain.)
CHMC_HD inline bool newton_select(const Work& w, int c, int& prev, int& qsel) {
  (void)prev, (void)qsel;
  return w.nw[c] == 1;
}

CHMC_HD inline unsigned long long dbits(double x) {
  union { double d; unsigned long long u; } c;
  c.d = x;
  return c.u;
}
CHMC_HD inline double bitsd(unsigned long long u) {
  union { double d; unsigned long long u; } c;
  c.u = u;
  return c.d;
}
// max over |x| as a bit pattern: non-negative doubles order like integers and NaN sorts above inf,
// which reproduces jnp.max(jnp.abs(.)) NaN propagation (:995-997)
CHMC_HD inline unsigned long long absbits(double x) { return dbits(fabs(x)) & 0x7fffffffffffffffULL; }

#if defined(__HIPCC__)
__device__ inline void atomic_max_u64(unsigned long long* a, unsigned long long v) { atomicMax(a, v); }
__device__ inline void atomic_add_i32(int* a, int v) { atomicAdd(a, v); }
__device__ inline void atomic_min_i32(int* a, int v) { atomicMin(a, v); }
__device__ inline void atomic_max_i32(int* a, int v) { atomicMax(a, v); }
#else
inline void atomic_min_i32(int* a, int v) {
  if (v < *a) *a = v;
}
inline void atomic_max_i32(int* a, int v) {
  if (v > *a) *a = v;
}
inline void atomic_max_u64(unsigned long long* a, unsigned long long v) {
  if (v > *a) *a = v;
}
inline void atomic_add_i32(int* a, int v) { *a += v; }
#endif

// ------------------------------------------------------------------------------------------
// small dense linear algebra on fixed-size (register) arrays
template <int N>
CHMC_HD inline double chol_lower(double* a) {  // in place, returns sum log |diag|
  double ld = 0.0;
  CHMC_UNROLL
  for (int j = 0; j < N; ++j) {
    double d = a[j * N + j];
    CHMC_UNROLL
    for (int k = 0; k < N; ++k)
      if (k < j) d -= a[j * N + k] * a[j * N + k];
    d = sqrt(d);
    a[j * N + j] = d;
    ld += log(fabs(d));
    double inv = 1.0 / d;
    CHMC_UNROLL
    for (int i = 0; i < N; ++i)
      if (i > j) {
        double t = a[i * N + j];
        CHMC_UNROLL
        for (int k = 0; k < N; ++k)
          if (k < j) t -= a[i * N + k] * a[j * N + k];
        a[i * N + j] = t * inv;
      }
    CHMC_UNROLL
    for (int i = 0; i < N; ++i)
      if (i < j) a[i * N + j] = 0.0;
  }
  return ld;
}
template <int N, int NR>
CHMC_HD inline void cho_solve(const double* L, double* b) {  // b [N][NR] in place
  CHMC_UNROLL
  for (int c = 0; c < NR; ++c) {
    CHMC_UNROLL
    for (int i = 0; i < N; ++i) {
      double t = b[i * NR + c];
      CHMC_UNROLL
      for (int k = 0; k < N; ++k)
        if (k < i) t -= L[i * N + k] * b[k * NR + c];
      b[i * NR + c] = t / L[i * N + i];
    }
    CHMC_UNROLL
    for (int i = N - 1; i >= 0; --i) {
      double t = b[i * NR + c];
      CHMC_UNROLL
      for (int k = 0; k < N; ++k)
        if (k > i) t -= L[k * N + i] * b[k * NR + c];
      b[i * NR + c] = t / L[i * N + i];
    }
  }
}
template <int N>
CHMC_HD inline void lu_factor(double* a, int* piv) {  // partial pivoting, LAPACK getrf semantics (:745-752)
  CHMC_UNROLL
  for (int j = 0; j < N; ++j) {
    int p = j;
    double mx = fabs(a[j * N + j]);
    CHMC_UNROLL
    for (int i = 0; i < N; ++i)
      if (i > j) {
        double v = fabs(a[i * N + j]);
        if (v > mx) mx = v, p = i;
      }
    piv[j] = p;
    CHMC_UNROLL
    for (int i = 0; i < N; ++i)
      if (i > j && i == p) {
        CHMC_UNROLL
        for (int k = 0; k < N; ++k) {
          double t = a[j * N + k];
          a[j * N + k] = a[i * N + k];
          a[i * N + k] = t;
        }
      }
    double inv = 1.0 / a[j * N + j];
    CHMC_UNROLL
    for (int i = 0; i < N; ++i)
      if (i > j) {
        double l = a[i * N + j] * inv;
        a[i * N + j] = l;
        CHMC_UNROLL
        for (int k = 0; k < N; ++k)
          if (k > j) a[i * N + k] -= l * a[j * N + k];
      }
  }
}
template <int N, int NR>
CHMC_HD inline void lu_solve(const double* lu, const int* piv, double* b) {  // b [N][NR] in place
  // (the row exchanges as selects on statically indexed entries: written as `if (t == p) swap`, hipcc turns the unrolled
  // comparison chain back into b[p], a dynamic index that puts the whole right-hand side into scratch memory -- every entry of
  // the substitutions below then goes through it: k_newton_fsm_wave, 360 scratch accesses per Newton iteration)
  CHMC_UNROLL
  for (int i = 0; i < N; ++i) {
    const int p = piv[i];
    CHMC_UNROLL
    for (int t = 0; t < N; ++t)
      if (t > i) {
        const bool sw = t == p;
        CHMC_UNROLL
        for (int c = 0; c < NR; ++c) {
          const double s = b[i * NR + c], u = b[t * NR + c];
          b[i * NR + c] = sw ? u : s;
          b[t * NR + c] = sw ? s : u;
        }
      }
  }
  CHMC_UNROLL
  for (int c = 0; c < NR; ++c) {
    CHMC_UNROLL
    for (int i = 0; i < N; ++i) {
      double t = b[i * NR + c];
      CHMC_UNROLL
      for (int k = 0; k < N; ++k)
        if (k < i) t -= lu[i * N + k] * b[k * NR + c];
      b[i * NR + c] = t;
    }
    CHMC_UNROLL
    for (int i = N - 1; i >= 0; --i) {
      double t = b[i * NR + c];
      CHMC_UNROLL
      for (int k = 0; k < N; ++k)
        if (k > i) t -= lu[i * N + k] * b[k * NR + c];
      b[i * NR + c] = t / lu[i * N + i];
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-(chain, block) scans
template <class M>
struct ChainConsts {
  double z[M::Z];
  double k[M::NK];
  CHMC_HD void init(const double* u, double dl) {
    M::gz(u, z);
    M::precompute(z, dl, k);
  }
};

// One block of `constr` (:473-519): generate_y_bar (:399-411) minus y_bar (:447-470).
// traj (may be null) receives nsteps+1 states; cp receives RM padded constraint values.
// The recursion is inherently sequential: what paces it is the dependent FMA chain of one step, so the loop is
// organised in tiles of 8 steps with (i) the noise increments of the next tile requested while the current one
// is integrated and (ii) no per-step branches when S is a multiple of 8 (bounds, trajectory store and the
// observation test are hoisted to tile level).
template <class M, int RM, bool STORE>
CHMC_HD inline void fwd_block_impl(const Sys& sy, const BlockDesc& bd, const ChainConsts<M>& cc, const double* q,
                                   const double* xobs, double* traj, double* cp) {
  constexpr int X = M::X, V = M::V, PF = 8;
  double x[X], xn[X];
  const double* vb = q + sy.U;
  if (bd.first) {
    M::gx0(cc.z, vb, x);
  } else {
    for (int a = 0; a < X; ++a) x[a] = xobs[(bd.obs0 - 1) * X + a];
  }
  const double* v = vb + sy.V0 + (size_t)bd.step0 * V;
  const double* n = q + sy.U + sy.NV;
  for (int i = 0; i < RM; ++i) cp[i] = 0.0;
  const int L = bd.nsteps, S = sy.S;
  const double sig = sy.noisy ? sigma_at(sy, q) : 0.0;
  // loads run up to PF steps past the block's end (never consumed): every q-like buffer is allocated with
  // CHMC_Q_PAD doubles of slack so that the scan needs no per-element bounds branches
  double cur[PF * V], nxt[PF * V];
  CHMC_UNROLL
  for (int i = 0; i < PF * V; ++i) cur[i] = v[i];
  if (S % PF == 0) {
    int left = S, j = 0;  // steps left until the next observation time
    for (int s0 = 0; s0 < L; s0 += PF) {
      const double* vn = v + (size_t)(s0 + PF) * V;
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) nxt[i] = vn[i];
      double tb[PF * X];
      CHMC_UNROLL
      for (int i = 0; i < PF; ++i) {
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) tb[i * X + a] = x[a];
        M::step(cc.k, x, cur + i * V, xn);
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) x[a] = xn[a];
      }
      if (STORE) {
        CHMC_UNROLL
        for (int i = 0; i < PF * X; ++i) traj[(size_t)s0 * X + i] = tb[i];
      }
      left -= PF;
      if (left == 0) {  // s0 + PF is the time of local observation j
        if (j < bd.ny) {
          double yv = M::obs(x);
          if (sy.noisy) yv += sig * n[bd.obs0 + j];
          cp[j] = yv - sy.y[bd.obs0 + j];
        }
        ++j;
        left = S;
      }
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) cur[i] = nxt[i];
    }
  } else {
    int cnt = S, j = 0;
    for (int s0 = 0; s0 < L; s0 += PF) {
      const double* vn = v + (size_t)(s0 + PF) * V;
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) nxt[i] = vn[i];
      CHMC_UNROLL
      for (int i = 0; i < PF; ++i) {
        const int s = s0 + i;
        if (s < L) {
          if (STORE)
            for (int a = 0; a < X; ++a) traj[(size_t)s * X + a] = x[a];
          M::step(cc.k, x, cur + i * V, xn);
          for (int a = 0; a < X; ++a) x[a] = xn[a];
          if (--cnt == 0) {  // s + 1 is the time of local observation j
            if (j < bd.ny) {
              double yv = M::obs(x);
              if (sy.noisy) yv += sig * n[bd.obs0 + j];
              cp[j] = yv - sy.y[bd.obs0 + j];
            }
            ++j;
            cnt = S;
          }
        }
      }
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) cur[i] = nxt[i];
    }
  }
  if (STORE)
    for (int a = 0; a < X; ++a) traj[(size_t)L * X + a] = x[a];
  if (!bd.last)
    for (int a = 0; a < X; ++a) cp[bd.ny + a] = x[a] - xobs[(bd.obs0 + bd.nobs - 1) * X + a];
}
template <class M, int RM>
CHMC_HD inline void fwd_block(const Sys& sy, const BlockDesc& bd, const ChainConsts<M>& cc, const double* q,
                              const double* xobs, double* traj, double* cp) {
  if (traj)
    fwd_block_impl<M, RM, true>(sy, bd, cc, q, xobs, traj, cp);
  else
    fwd_block_impl<M, RM, false>(sy, bd, cc, q, xobs, traj, cp);
}

// Reverse (adjoint) sweep over a block carrying all RM constraint rows (jacob_constr_blocks :521-624).
// MODE 0: store dc/dv rows into Jv_out and accumulate the symmetric Gram block D = Jv Jv^T (:765-792).
// MODE 1: do not store; accumulate D = Jv(q) Jv(q_prev)^T against the stored rows Jr (:742-744).
// JuL receives the RM x U rows of dc/du.  D is RM x RM, identity-padded, noise term added.
template <class M, int RM, int MODE>
CHMC_HD inline void rev_block(const Sys& sy, const BlockDesc& bd, const ChainConsts<M>& cc, const double* u,
                              const double* q, const double* traj, double* Jv_out, const double* Jr, double* JuL,
                              double* D, double sig_prev) {  // sig_prev: sigma at the previous point (MODE 1)
  constexpr int X = M::X, V = M::V, Z = M::Z, U = M::U;
  const int S = sy.S, L = bd.nsteps, NV = sy.NV;
  double Lam[RM * X], zbar[RM * Z];
  for (int i = 0; i < RM * X; ++i) Lam[i] = 0.0;
  for (int i = 0; i < RM * Z; ++i) zbar[i] = 0.0;
  for (int i = 0; i < RM * RM; ++i) D[i] = 0.0;
  const double* v = q + sy.U + sy.V0 + (size_t)bd.step0 * V;
  const size_t colb = (size_t)sy.V0 + (size_t)bd.step0 * V;
  int next_obs = bd.nobs;  // observation whose time is the end of the current interval
  int cnt = 0;             // steps until that observation time is crossed
  for (int st = L - 1; st >= 0; --st) {
    if (cnt == 0) {  // st + 1 == next_obs * S
      const int j = next_obs - 1;
      if (j < bd.ny) {
        double g[X];
        M::obs_grad(traj + (size_t)(st + 1) * X, g);
        CHMC_UNROLL
        for (int i = 0; i < RM; ++i)
          if (i == j)
            for (int a = 0; a < X; ++a) Lam[i * X + a] = g[a];
      }
      if (st + 1 == L && !bd.last) {
        CHMC_UNROLL
        for (int i = 0; i < RM; ++i)
          for (int a = 0; a < X; ++a)
            if (i == bd.ny + a) Lam[i * X + a] = 1.0;
      }
      --next_obs;
      cnt = S;
    }
    --cnt;
    double A[X * X], Bm[X * V], Zf[X * Z];
    M::jac(cc.k, traj + (size_t)st * X, v + (size_t)st * V, A, Bm, Zf);
    double jr[RM * V];
    CHMC_UNROLL
    for (int i = 0; i < RM; ++i) {
      CHMC_UNROLL
      for (int c = 0; c < V; ++c) {
        double t = 0.0;
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) t += Lam[i * X + a] * Bm[a * V + c];
        jr[i * V + c] = t;
      }
    }
    const size_t col = colb + (size_t)st * V;
    if (MODE == 0) {
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        CHMC_UNROLL
        for (int c = 0; c < V; ++c) Jv_out[(size_t)i * NV + col + c] = jr[i * V + c];
      }
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        CHMC_UNROLL
        for (int j = 0; j <= i; ++j) {
          double t = D[i * RM + j];
          CHMC_UNROLL
          for (int c = 0; c < V; ++c) t += jr[i * V + c] * jr[j * V + c];
          D[i * RM + j] = t;
        }
      }
    } else {
      double jp[RM * V];
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        CHMC_UNROLL
        for (int c = 0; c < V; ++c) jp[i * V + c] = Jr[(size_t)i * NV + col + c];
      }
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        CHMC_UNROLL
        for (int j = 0; j < RM; ++j) {
          double t = D[i * RM + j];
          CHMC_UNROLL
          for (int c = 0; c < V; ++c) t += jr[i * V + c] * jp[j * V + c];
          D[i * RM + j] = t;
        }
      }
    }
    CHMC_UNROLL
    for (int i = 0; i < RM; ++i) {
      CHMC_UNROLL
      for (int mz = 0; mz < Z; ++mz) {
        double t = zbar[i * Z + mz];
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) t += Lam[i * X + a] * Zf[a * Z + mz];
        zbar[i * Z + mz] = t;
      }
      double nl[X];
      CHMC_UNROLL
      for (int c = 0; c < X; ++c) {
        double t = 0.0;
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) t += Lam[i * X + a] * A[a * X + c];
        nl[c] = t;
      }
      CHMC_UNROLL
      for (int c = 0; c < X; ++c) Lam[i * X + c] = nl[c];
    }
  }
  if (bd.first) {  // x_0 = generate_x_0(z, v_0): columns of v_0 and the z-dependence (:401, :566-567)
    double dz[X * Z], dv0[X * M::V0];
    M::gx0_jac(dz, dv0);
    double j0[RM * M::V0];
    for (int i = 0; i < RM; ++i) {
      for (int c = 0; c < M::V0; ++c) {
        double t = 0.0;
        for (int a = 0; a < X; ++a) t += Lam[i * X + a] * dv0[a * M::V0 + c];
        j0[i * M::V0 + c] = t;
      }
      for (int mz = 0; mz < Z; ++mz) {
        double t = 0.0;
        for (int a = 0; a < X; ++a) t += Lam[i * X + a] * dz[a * Z + mz];
        zbar[i * Z + mz] += t;
      }
    }
    if (MODE == 0) {
      for (int i = 0; i < RM; ++i)
        for (int c = 0; c < M::V0; ++c) Jv_out[(size_t)i * NV + c] = j0[i * M::V0 + c];
      for (int i = 0; i < RM; ++i)
        for (int j = 0; j <= i; ++j)
          for (int c = 0; c < M::V0; ++c) D[i * RM + j] += j0[i * M::V0 + c] * j0[j * M::V0 + c];
    } else {
      for (int i = 0; i < RM; ++i)
        for (int j = 0; j < RM; ++j)
          for (int c = 0; c < M::V0; ++c) D[i * RM + j] += j0[i * M::V0 + c] * Jr[(size_t)j * NV + c];
    }
  }
  if (MODE == 0) {
    for (int i = 0; i < RM; ++i)
      for (int j = 0; j < i; ++j) D[j * RM + i] = D[i * RM + j];
  }
  const double sig = sy.noisy ? sigma_at(sy, q) : 0.0;
  const double s2 = sig * (MODE == 0 ? sig : sig_prev);  // dc_dn_l * dc_dn_r (:772-791)
  for (int i = 0; i < RM; ++i) {
    if (sy.noisy && i < bd.ny) D[i * RM + i] += s2;  // dc/dn dc/dn^T (:772-791)
    if (i >= bd.nrows) D[i * RM + i] = 1.0;          // identity padding
  }
  double G[Z * Z];
  M::gz_jac(u, G);
  for (int i = 0; i < RM; ++i) {
    for (int c = 0; c < Z; ++c) {
      double t = 0.0;
      for (int mz = 0; mz < Z; ++mz) t += zbar[i * Z + mz] * G[mz * Z + c];
      JuL[i * U + c] = t;
    }
    // d(sigma(u) n_i) / du_sigma = sigma n_i on the observation rows (g_y_bar :559-569)
    if (M::VS) JuL[i * U + Z] = i < bd.ny ? sig * q[sy.U + sy.NV + bd.obs0 + i] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------
// kernels (functors)
#define CHMC_CB_DECODE                   \
  const int c = tid / sy.K;              \
  const int b = tid - c * sy.K;          \
  const BlockDesc bd = sy.blk[b];

// constr only (quasi-Newton iterations, per-op API).  qsel: 0 slot `which`, 1 work.qb
template <class M, int RM>
struct KFwd {
  Sys sy;
  Slots sl;
  Work w;
  int which, qsel, use_nw, store_traj;
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    int which = this->which, qsel = this->qsel;
    if (use_nw ? w.nw[c] != 1 : !w.ok[c]) return;
    const int s = sl.cur[c] ^ which;
    const double* q = (qsel ? w.qb : pick(sl.q, s)) + (size_t)c * sy.Q;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    // store_traj: 0 none, 1 into the slot, 2 into the Newton-iterate work trajectory
    double* traj = store_traj ? (store_traj == 2 ? w.trajw : pick(sl.traj, s)) + (size_t)c * sy.TRJ + (size_t)(bd.step0 + CHMC_TPAD * b) * M::X
                              : nullptr;
    double cp[RM];
    fwd_block<M, RM>(sy, bd, cc, q, sy.xobs + (size_t)c * sy.T * M::X, traj, cp);
    double* out = w.cpad + ((size_t)c * sy.Kmax + b) * RM;
    for (int i = 0; i < RM; ++i) out[i] = cp[i];
  }
};

// generate_x_obs_seq (:384-397): one work item per chain; same software-prefetched scan as fwd_block
template <class M>
struct KXobs {
  Sys sy;
  Slots sl;
  double* xobs_out;
  CHMC_HD void operator()(int c) const {
    constexpr int X = M::X, V = M::V, PF = 8;
    const double* q = pick(sl.q, sl.cur[c]) + (size_t)c * sy.Q;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    double x[X], xn[X];
    M::gx0(cc.z, q + sy.U, x);
    const double* v = q + sy.U + sy.V0;
    double* out = xobs_out + (size_t)c * sy.T * X;
    const int L = sy.T * sy.S;
    if (sy.S % PF == 0) {  // same ring of in-flight tiles as fwd_block_impl
      constexpr int DEPTH = CHMC_FWD_DEPTH;
      double ring[DEPTH][PF * V];
      CHMC_UNROLL
      for (int d = 0; d < DEPTH; ++d) {
        CHMC_UNROLL
        for (int i = 0; i < PF * V; ++i) ring[d][i] = v[d * PF * V + i];
      }
      int left = sy.S, t = 0;
      for (int s0 = 0; s0 < L; s0 += PF * DEPTH) {
        CHMC_UNROLL
        for (int d = 0; d < DEPTH; ++d) {
          const int s = s0 + d * PF;
          if (s < L) {
            CHMC_UNROLL
            for (int i = 0; i < PF; ++i) {
              M::step(cc.k, x, ring[d] + i * V, xn);
              CHMC_UNROLL
              for (int a = 0; a < X; ++a) x[a] = xn[a];
            }
            const double* vn = v + (size_t)(s + PF * DEPTH) * V;
            CHMC_UNROLL
            for (int i = 0; i < PF * V; ++i) ring[d][i] = vn[i];
            left -= PF;
            if (left == 0) {
              for (int a = 0; a < X; ++a) out[t * X + a] = x[a];
              ++t;
              left = sy.S;
            }
          }
        }
      }
      return;
    }
    double cur[PF * V], nxt[PF * V];
    CHMC_UNROLL
    for (int i = 0; i < PF * V; ++i) cur[i] = v[i];
    int cnt = sy.S, t = 0;
    for (int s0 = 0; s0 < L; s0 += PF) {
      const double* vn = v + (size_t)(s0 + PF) * V;
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) nxt[i] = vn[i];
      CHMC_UNROLL
      for (int i = 0; i < PF; ++i) {
        if (s0 + i < L) {
          M::step(cc.k, x, cur + i * V, xn);
          for (int a = 0; a < X; ++a) x[a] = xn[a];
          if (--cnt == 0) {
            for (int a = 0; a < X; ++a) out[t * X + a] = x[a];
            ++t;
            cnt = sy.S;
          }
        }
      }
      CHMC_UNROLL
      for (int i = 0; i < PF * V; ++i) cur[i] = nxt[i];
    }
  }
};

// Forward half of conditioned_diffusion_neg_log_dens_and_grad (:82-205, the unconstrained-HMC comparator of the
// reference's experiments) when the steps per observation do not tile by 8 (otherwise k_fwd_scan runs the whole
// chain as one block): the full T * S-step scan of one chain from q = [u | v_0 | v_seq], storing the trajectory.
template <class M>
struct KFullScan {
  Sys sy;
  const double* qin;  // [B][QH], QH = U + V0 + T S V
  double* traj;       // [B][TRJ]
  int QH;
  CHMC_HD void operator()(int c) const {
    constexpr int X = M::X, V = M::V;
    const double* q = qin + (size_t)c * QH;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    double x[X], xn[X];
    M::gx0(cc.z, q + sy.U, x);
    const double* v = q + sy.U + sy.V0;
    double* tr = traj + (size_t)c * sy.TRJ;
    const int L = sy.T * sy.S;
    for (int s = 0; s < L; ++s) {
      for (int a = 0; a < X; ++a) tr[(size_t)s * X + a] = x[a];
      M::step(cc.k, x, v + (size_t)s * V, xn);
      for (int a = 0; a < X; ++a) x[a] = xn[a];
    }
    for (int a = 0; a < X; ++a) tr[(size_t)L * X + a] = x[a];
  }
};

// State evaluation, block part: trajectory, Jacobian rows, Gram block, its Cholesky factor,
// D^-1 dc/du and this block's contribution to C (jacob_constr_blocks + chol_gram_blocks :626-687)
template <class M, int RM>
struct KStateBlk {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    if (!w.ok[c]) return;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    const double* q = pick(sl.q, s) + (size_t)c * sy.Q;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    double* traj = pick(sl.traj, s) + (size_t)c * sy.TRJ + (size_t)(bd.step0 + CHMC_TPAD * b) * M::X;
    double cp[RM];
    fwd_block<M, RM>(sy, bd, cc, q, sy.xobs + (size_t)c * sy.T * M::X, traj, cp);
    double* cout_ = w.cpad + ((size_t)c * sy.Kmax + b) * RM;
    for (int i = 0; i < RM; ++i) cout_[i] = cp[i];
    double D[RM * RM], Ju[RM * U];
    rev_block<M, RM, 0>(sy, bd, cc, q, q, traj, pick(sl.Jv, s) + (size_t)c * RM * sy.NV, nullptr, Ju, D, 0.0);
    const size_t cb = (size_t)c * sy.Kmax + b;
    double ld = chol_lower<RM>(D);
    double* fd = pick(sl.facD, s) + cb * RM * RM;
    for (int i = 0; i < RM * RM; ++i) fd[i] = D[i];
    pick(sl.ldb, s)[cb] = ld;
    double* ju = pick(sl.JuP, s) + cb * RM * U;
    for (int i = 0; i < RM * U; ++i) ju[i] = Ju[i];
    double E[RM * U];
    for (int i = 0; i < RM * U; ++i) E[i] = Ju[i];
    cho_solve<RM, U>(D, E);
    double* eo = pick(sl.E, s) + cb * RM * U;
    for (int i = 0; i < RM * U; ++i) eo[i] = E[i];
    double* Cb = w.Cb + cb * U * U;
    for (int a = 0; a < U; ++a)
      for (int d = 0; d < U; ++d) {
        double t = 0.0;
        for (int i = 0; i < RM; ++i) t += Ju[i * U + a] * E[i * U + d];
        Cb[a * U + d] = t;
      }
  }
};

// State evaluation, chain part: C = M_0 + sum_b Ju_b^T D_b^-1 Ju_b, its Cholesky factor and inverse,
// 1/2 log det Gram (:676-686, :800-810)
template <class M>
struct KStateChain {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_FI CHMC_HD void operator()(int c) const {
    if (!w.ok[c]) return;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    double Cm[U * U];
    for (int i = 0; i < U * U; ++i) Cm[i] = 0.0;
    for (int i = 0; i < U; ++i) Cm[i * U + i] = 1.0;
    if (sy.m0)  // get_M_0_matrix (:794-798)
      for (int i = 0; i < U * U; ++i) Cm[i] = sy.m0[i];
    double ld = sy.m0 ? -sy.hld_m0 : 0.0;  // - log_det_sqrt_metric_0 (:809)
    for (int b = 0; b < sy.K; ++b) {
      const double* Cb = w.Cb + ((size_t)c * sy.Kmax + b) * U * U;
      for (int i = 0; i < U * U; ++i) Cm[i] += Cb[i];
      ld += pick(sl.ldb, s)[(size_t)c * sy.Kmax + b];
    }
    ld += chol_lower<U>(Cm);
    double Ci[U * U];
    for (int i = 0; i < U * U; ++i) Ci[i] = 0.0;
    for (int i = 0; i < U; ++i) Ci[i * U + i] = 1.0;
    cho_solve<U, U>(Cm, Ci);
    for (int i = 0; i < U * U; ++i) {
      pick(sl.facC, s)[(size_t)c * U * U + i] = Cm[i];
      pick(sl.Cinv, s)[(size_t)c * U * U + i] = Ci[i];
    }
    pick(sl.logdet, s)[c] = ld;
  }
};

// Variable observation noise, sigma = generate_sigma(u) = exp(u_sigma): on the observation rows
// c_i = obs_func(x) + sigma(u) n_i - y_i, so the Jacobian entries dc_i/du_sigma = sigma n_i and dc_i/dn_i = sigma depend
// on (u_sigma, n_i).  With W = G^-1 J M^-1 (W[i, u_sigma] = Wu[i][Z], W[i, n_i] = (G^-1)_ii sigma) the gradient of
// 1/2 log det G gains   d/du_sigma: sum_i sigma (Wu[i][Z] n_i + sigma (G^-1)_ii)   and   d/dn_i: sigma Wu[i][Z].
// Writes the n-components of this block into g (the chain's gradient) and returns the block's u_sigma term.
template <int RM>
CHMC_HD inline double var_sigma_grad_terms(const Sys& sy, const BlockDesc& bd, const double* q, const double* Wu, int U,
                                           const double* Mb, double* g) {
  const double sg = sigma_at(sy, q);
  const double* n = q + sy.U + sy.NV + bd.obs0;
  double* gn = g + sy.U + sy.NV + bd.obs0;
  double acc = 0.0;
  for (int i = 0; i < RM; ++i)
    if (i < bd.ny) {
      const double wz = Wu[i * U + sy.Z];
      acc += sg * (wz * n[i] + sg * Mb[i * RM + i]);
      gn[i] = sg * wz;
    }
  return acc;
}

// Gradient of 1/2 log det Gram, block part (value_and_grad of log_det_sqrt_gram :812-820, :1143-1146):
// grad = sum_i grad_q [ grad c_i . w_i ],  W = G^-1 J on J's block pattern, i.e. per block one tangent
// sweep per row along w_i and one second-order adjoint sweep with the sources summed over rows.
template <class M, int RM>
struct KGldBlk {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    if (!w.ok[c]) return;
    constexpr int X = M::X, V = M::V, Z = M::Z, U = M::U, V0 = M::V0, NXI = M::NXI;
    const int s = sl.cur[c] ^ which;
    const int S = sy.S, L = bd.nsteps, NV = sy.NV;
    const size_t cb = (size_t)c * sy.Kmax + b;
    const double* q = pick(sl.q, s) + (size_t)c * sy.Q;
    const double* traj = pick(sl.traj, s) + (size_t)c * sy.TRJ + (size_t)(bd.step0 + CHMC_TPAD * b) * X;
    const double* Jv = pick(sl.Jv, s) + (size_t)c * RM * NV;
    const double* v = q + sy.U + sy.V0 + (size_t)bd.step0 * V;
    const size_t colb = (size_t)sy.V0 + (size_t)bd.step0 * V;
    double* gv = pick(sl.grad, s) + (size_t)c * sy.Q + sy.U;
    double* Xd = w.Xd + ((size_t)c * sy.T * S + bd.step0) * RM * X;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    double Gz[Z * Z], dx0dz[X * Z], dx0dv0[X * V0];
    M::gz_jac(q, Gz);
    M::gx0_jac(dx0dz, dx0dv0);
    // Wu = D^-1 Ju C^-1 (rows of G^-1 dc/du);  Mb = (G^-1)_bb = D^-1 - E C^-1 E^T
    double Wu[RM * U], Mb[RM * RM], zd[RM * Z];
    {
      const double* E = pick(sl.E, s) + cb * RM * U;
      const double* Ci = pick(sl.Cinv, s) + (size_t)c * U * U;
      double Dl[RM * RM];
      const double* fd = pick(sl.facD, s) + cb * RM * RM;
      for (int i = 0; i < RM * RM; ++i) Dl[i] = fd[i];
      for (int i = 0; i < RM; ++i)
        for (int d = 0; d < U; ++d) {
          double t = 0.0;
          for (int a = 0; a < U; ++a) t += E[i * U + a] * Ci[a * U + d];
          Wu[i * U + d] = t;
        }
      for (int i = 0; i < RM * RM; ++i) Mb[i] = 0.0;
      for (int i = 0; i < RM; ++i) Mb[i * RM + i] = 1.0;
      cho_solve<RM, RM>(Dl, Mb);
      for (int i = 0; i < RM; ++i)
        for (int j = 0; j < RM; ++j) {
          double t = 0.0;
          for (int a = 0; a < U; ++a) t += Wu[i * U + a] * E[j * U + a];
          Mb[i * RM + j] -= t;
        }
      for (int i = 0; i < RM; ++i) {
        if (i >= bd.nrows)  // padded rows carry no constraint
          for (int j = 0; j < RM; ++j) Mb[i * RM + j] = 0.0, Mb[j * RM + i] = 0.0;
        for (int mz = 0; mz < Z; ++mz) {
          double t = 0.0;
          for (int d = 0; d < Z; ++d) t += Gz[mz * Z + d] * Wu[i * U + d];
          zd[i * Z + mz] = t;
        }
      }
    }
    // forward tangent sweeps
    double xd[RM * X], xdt[RM * X];
    for (int i = 0; i < RM * X; ++i) xd[i] = 0.0, xdt[i] = 0.0;
    if (bd.first)
      for (int i = 0; i < RM; ++i)
        for (int a = 0; a < X; ++a) {
          double t = 0.0;
          for (int mz = 0; mz < Z; ++mz) t += dx0dz[a * Z + mz] * zd[i * Z + mz];
          for (int d = 0; d < V0; ++d) {
            double wv = 0.0;
            for (int j = 0; j < RM; ++j) wv += Mb[i * RM + j] * Jv[(size_t)j * NV + d];
            t += dx0dv0[a * V0 + d] * wv;
          }
          xd[i * X + a] = t;
        }
    {
      int cnt = S, jobs = 0;
      for (int st = 0; st < L; ++st) {
        for (int i = 0; i < RM * X; ++i) Xd[(size_t)st * RM * X + i] = xd[i];
        double A[X * X], Bm[X * V], Zf[X * Z], jp[RM * V];
        M::jac(cc.k, traj + (size_t)st * X, v + (size_t)st * V, A, Bm, Zf);
        const size_t col = colb + (size_t)st * V;
        CHMC_UNROLL
        for (int j = 0; j < RM; ++j) {
          CHMC_UNROLL
          for (int d = 0; d < V; ++d) jp[j * V + d] = Jv[(size_t)j * NV + col + d];
        }
        CHMC_UNROLL
        for (int i = 0; i < RM; ++i) {
          double wv[V], nx[X];
          CHMC_UNROLL
          for (int d = 0; d < V; ++d) {
            double t = 0.0;
            CHMC_UNROLL
            for (int j = 0; j < RM; ++j) t += Mb[i * RM + j] * jp[j * V + d];
            wv[d] = t;
          }
          CHMC_UNROLL
          for (int a = 0; a < X; ++a) {
            double t = 0.0;
            CHMC_UNROLL
            for (int d = 0; d < X; ++d) t += A[a * X + d] * xd[i * X + d];
            CHMC_UNROLL
            for (int d = 0; d < V; ++d) t += Bm[a * V + d] * wv[d];
            CHMC_UNROLL
            for (int mz = 0; mz < Z; ++mz) t += Zf[a * Z + mz] * zd[i * Z + mz];
            nx[a] = t;
          }
          CHMC_UNROLL
          for (int a = 0; a < X; ++a) xd[i * X + a] = nx[a];
        }
        if (--cnt == 0) {  // st + 1 is the time of local observation `jobs`
          if (jobs < bd.ny) {
            CHMC_UNROLL
            for (int i = 0; i < RM; ++i)
              if (i == jobs)
                for (int a = 0; a < X; ++a) xdt[i * X + a] = xd[i * X + a];
          }
          ++jobs;
          cnt = S;
        }
      }
    }
    // backward second-order sweep
    double Lam[RM * X], zbar[RM * Z], xbar[X], zbt[Z];
    for (int i = 0; i < RM * X; ++i) Lam[i] = 0.0;
    for (int i = 0; i < RM * Z; ++i) zbar[i] = 0.0;
    for (int a = 0; a < X; ++a) xbar[a] = 0.0;
    for (int a = 0; a < Z; ++a) zbt[a] = 0.0;
    int next_obs = bd.nobs, cnt = 0;
    for (int st = L - 1; st >= 0; --st) {
      if (cnt == 0) {
        const int j = next_obs - 1;
        if (j < bd.ny) {
          double g[X], hv[X], xt[X];
          CHMC_UNROLL
          for (int i = 0; i < RM; ++i)
            if (i == j)
              for (int a = 0; a < X; ++a) xt[a] = xdt[i * X + a];
          M::obs_grad(traj + (size_t)(st + 1) * X, g);
          M::obs_hess_vec(traj + (size_t)(st + 1) * X, xt, hv);
          CHMC_UNROLL
          for (int i = 0; i < RM; ++i)
            if (i == j)
              for (int a = 0; a < X; ++a) Lam[i * X + a] = g[a];
          for (int a = 0; a < X; ++a) xbar[a] += hv[a];
        }
        if (st + 1 == L && !bd.last) {
          CHMC_UNROLL
          for (int i = 0; i < RM; ++i)
            for (int a = 0; a < X; ++a)
              if (i == bd.ny + a) Lam[i * X + a] = 1.0;
        }
        --next_obs;
        cnt = S;
      }
      --cnt;
      double A[X * X], Bm[X * V], Zf[X * Z], jp[RM * V], Sm[X * NXI], H[NXI];
      M::jac(cc.k, traj + (size_t)st * X, v + (size_t)st * V, A, Bm, Zf);
      const size_t col = colb + (size_t)st * V;
      CHMC_UNROLL
      for (int j = 0; j < RM; ++j) {
        CHMC_UNROLL
        for (int d = 0; d < V; ++d) jp[j * V + d] = Jv[(size_t)j * NV + col + d];
      }
      for (int i = 0; i < X * NXI; ++i) Sm[i] = 0.0;
      const double* xds = Xd + (size_t)st * RM * X;
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        double dir[NXI];
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) dir[a] = xds[i * X + a];
        CHMC_UNROLL
        for (int d = 0; d < V; ++d) {
          double t = 0.0;
          CHMC_UNROLL
          for (int j = 0; j < RM; ++j) t += Mb[i * RM + j] * jp[j * V + d];
          dir[X + d] = t;
        }
        CHMC_UNROLL
        for (int mz = 0; mz < Z; ++mz) dir[X + V + mz] = zd[i * Z + mz];
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) {
          CHMC_UNROLL
          for (int m2 = 0; m2 < NXI; ++m2) Sm[a * NXI + m2] += Lam[i * X + a] * dir[m2];
        }
      }
      M::hess(cc.k, traj + (size_t)st * X, v + (size_t)st * V, Sm, H);
      for (int d = 0; d < V; ++d) {
        double t = H[X + d];
        for (int a = 0; a < X; ++a) t += Bm[a * V + d] * xbar[a];
        gv[col + d] = t;
      }
      for (int mz = 0; mz < Z; ++mz) {
        double t = H[X + V + mz];
        for (int a = 0; a < X; ++a) t += Zf[a * Z + mz] * xbar[a];
        zbt[mz] += t;
      }
      double nxb[X];
      for (int d = 0; d < X; ++d) {
        double t = H[d];
        for (int a = 0; a < X; ++a) t += A[a * X + d] * xbar[a];
        nxb[d] = t;
      }
      for (int d = 0; d < X; ++d) xbar[d] = nxb[d];
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        CHMC_UNROLL
        for (int mz = 0; mz < Z; ++mz) {
          double t = zbar[i * Z + mz];
          CHMC_UNROLL
          for (int a = 0; a < X; ++a) t += Lam[i * X + a] * Zf[a * Z + mz];
          zbar[i * Z + mz] = t;
        }
        double nl[X];
        CHMC_UNROLL
        for (int d = 0; d < X; ++d) {
          double t = 0.0;
          CHMC_UNROLL
          for (int a = 0; a < X; ++a) t += Lam[i * X + a] * A[a * X + d];
          nl[d] = t;
        }
        CHMC_UNROLL
        for (int d = 0; d < X; ++d) Lam[i * X + d] = nl[d];
      }
    }
    if (bd.first) {
      for (int d = 0; d < V0; ++d) {
        double t = 0.0;
        for (int a = 0; a < X; ++a) t += dx0dv0[a * V0 + d] * xbar[a];
        gv[d] = t;
      }
      for (int mz = 0; mz < Z; ++mz) {
        double t = 0.0;
        for (int a = 0; a < X; ++a) t += dx0dz[a * Z + mz] * xbar[a];
        zbt[mz] += t;
      }
      for (int i = 0; i < RM; ++i)
        for (int mz = 0; mz < Z; ++mz) {
          double t = 0.0;
          for (int a = 0; a < X; ++a) t += Lam[i * X + a] * dx0dz[a * Z + mz];
          zbar[i * Z + mz] += t;
        }
    }
    double gu[U];
    for (int d = 0; d < Z; ++d) {
      double t = 0.0;
      for (int mz = 0; mz < Z; ++mz) t += Gz[mz * Z + d] * zbt[mz];
      gu[d] = t;
    }
    for (int i = 0; i < RM; ++i) {
      double o[Z];
      M::gz_hess(q, Wu + i * U, zbar + i * Z, o);
      for (int d = 0; d < Z; ++d) gu[d] += o[d];
    }
    if constexpr (M::VS) gu[Z] = var_sigma_grad_terms<RM>(sy, bd, q, Wu, U, Mb, pick(sl.grad, s) + (size_t)c * sy.Q);
    for (int d = 0; d < U; ++d) w.gup[cb * U + d] = gu[d];
  }
};

// Per-block quantities of the grad-log-det evaluation that do not depend on the time step (wave kernels read them):
// Wu = D^-1 Ju C^-1 (rows of G^-1 dc/du), Mb = (G^-1)_bb = D^-1 - E C^-1 E^T, zd_i = generate_z'(u) Wu_i.
template <class M, int RM>
struct KGldPrep {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    if (!w.ok[c]) return;
    constexpr int Z = M::Z, U = M::U;
    const int s = sl.cur[c] ^ which;
    const size_t cb = (size_t)c * sy.Kmax + b;
    const double* q = pick(sl.q, s) + (size_t)c * sy.Q;
    double Gz[Z * Z], Wu[RM * U], Mb[RM * RM], Dl[RM * RM];
    M::gz_jac(q, Gz);
    const double* E = pick(sl.E, s) + cb * RM * U;
    const double* Ci = pick(sl.Cinv, s) + (size_t)c * U * U;
    for (int i = 0; i < RM * RM; ++i) Dl[i] = pick(sl.facD, s)[cb * RM * RM + i];
    for (int i = 0; i < RM; ++i)
      for (int d = 0; d < U; ++d) {
        double t = 0.0;
        for (int a = 0; a < U; ++a) t += E[i * U + a] * Ci[a * U + d];
        Wu[i * U + d] = t;
      }
    for (int i = 0; i < RM * RM; ++i) Mb[i] = 0.0;
    for (int i = 0; i < RM; ++i) Mb[i * RM + i] = 1.0;
    cho_solve<RM, RM>(Dl, Mb);
    for (int i = 0; i < RM; ++i)
      for (int j = 0; j < RM; ++j) {
        double t = 0.0;
        for (int a = 0; a < U; ++a) t += Wu[i * U + a] * E[j * U + a];
        Mb[i * RM + j] -= t;
      }
    for (int i = 0; i < RM; ++i) {
      if (i >= bd.nrows)
        for (int j = 0; j < RM; ++j) Mb[i * RM + j] = 0.0, Mb[j * RM + i] = 0.0;
    }
    for (int i = 0; i < RM * RM; ++i) w.gMb[cb * RM * RM + i] = Mb[i];
    for (int i = 0; i < RM; ++i) {
      for (int d = 0; d < U; ++d) w.gWu[(cb * RM + i) * U + d] = Wu[i * U + d];
      for (int mz = 0; mz < Z; ++mz) {
        double t = 0.0;
        for (int d = 0; d < Z; ++d) t += Gz[mz * Z + d] * Wu[i * U + d];
        w.gzd[(cb * RM + i) * Z + mz] = t;
      }
    }
  }
};

template <class M>
struct KGldChain {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_FI CHMC_HD void operator()(int c) const {
    if (!w.ok[c]) return;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    double* g = pick(sl.grad, s) + (size_t)c * sy.Q;
    double t[U];  // (every component summed over the blocks in ascending order, the loads of four blocks in flight together:
                  // 19.8 -> 9.2 us per launch; the same unrolling of KStateChain's sum changes nothing, its time is the U x U inverse)
    CHMC_UNROLL
    for (int d = 0; d < U; ++d) t[d] = 0.0;
    CHMC_UNROLL4
    for (int b = 0; b < sy.K; ++b) {
      const double* gb = w.gup + ((size_t)c * sy.Kmax + b) * U;
      CHMC_UNROLL
      for (int d = 0; d < U; ++d) t[d] += gb[d];
    }
    CHMC_UNROLL
    for (int d = 0; d < U; ++d) g[d] = t[d];
    if (sy.noisy && !sy.varsig)  // fixed observation noise: the Gram matrix does not depend on n
      for (int t = 0; t < sy.T; ++t) g[sy.U + sy.NV + t] = 0.0;  // (variable noise: written by var_sigma_grad_terms)
  }
};

// Newton iteration, block part (body of newton_projection :1088-1104): constraint, Jacobian of the
// iterate contracted on the fly with the stored Jacobian of the previous point, LU of the block,
// D^-1 c, D^-1 dc/du and this block's share of C and of dc/du_prev^T D^-1 c.
template <class M, int RM>
struct KNewtonBlk {
  Sys sy;
  Slots sl;
  Work w;
  int prev, qsel;  // prev: slot (0/1) holding J(q_prev); qsel: 0 iterate = other slot's q, 1 iterate = work.qb
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    int prev = this->prev, qsel = this->qsel;
    if (!newton_select(w, c, prev, qsel)) return;
    constexpr int U = M::U;
    const int sp = sl.cur[c] ^ prev;
    const double* q = (qsel ? w.qb : pick(sl.q, sp ^ 1)) + (size_t)c * sy.Q;
    ChainConsts<M> cc;
    cc.init(q, sy.dl);
    double* traj = w.trajw + (size_t)c * sy.TRJ + (size_t)(bd.step0 + CHMC_TPAD * b) * M::X;
    double cp[RM];
    fwd_block<M, RM>(sy, bd, cc, q, sy.xobs + (size_t)c * sy.T * M::X, traj, cp);
    const size_t cb = (size_t)c * sy.Kmax + b;
    double* cout_ = w.cpad + cb * RM;
    for (int i = 0; i < RM; ++i) cout_[i] = cp[i];
    double D[RM * RM], JuL[RM * U];
    rev_block<M, RM, 1>(sy, bd, cc, q, q, traj, nullptr, pick(sl.Jv, sp) + (size_t)c * RM * sy.NV, JuL, D,
                        sy.noisy ? sigma_at(sy, pick(sl.q, sp) + (size_t)c * sy.Q) : 0.0);
    int piv[RM];
    lu_factor<RM>(D, piv);
    lu_solve<RM, 1>(D, piv, cp);
    lu_solve<RM, U>(D, piv, JuL);
    double* to = w.tpad + cb * RM;
    for (int i = 0; i < RM; ++i) to[i] = cp[i];
    double* eo = w.Ew + cb * RM * U;
    for (int i = 0; i < RM * U; ++i) eo[i] = JuL[i];
    const double* jur = pick(sl.JuP, sp) + cb * RM * U;
    double* Cb = w.Cb + cb * U * U;
    double* sb = w.sb + cb * U;
    for (int a = 0; a < U; ++a) {
      double t2 = 0.0;
      for (int i = 0; i < RM; ++i) t2 += jur[i * U + a] * cp[i];
      sb[a] = t2;
      for (int d = 0; d < U; ++d) {
        double t = 0.0;
        for (int i = 0; i < RM; ++i) t += jur[i * U + a] * JuL[i * U + d];
        Cb[a * U + d] = t;
      }
    }
  }
};

// Factor kernels: the small dense algebra of one block, reading the Gram block D and the dc/du rows produced
// by the wave-level reverse sweep (chmc_wave.h).  One work item per (chain, block) so that all 64 lanes of a
// wave factor 64 different blocks.
template <class M, int RM>
struct KNewtonFactor {  // LU of D = Jv(q) Jv(q_prev)^T + diag, D^-1 c, D^-1 dc/du, C_b, s_b  (:745-762, :957-969)
  Sys sy;
  Slots sl;
  Work w;
  int prev;
  CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    int prev = this->prev, qsel_unused = 0;
    if (!newton_select(w, c, prev, qsel_unused)) return;
    (void)bd;
    constexpr int U = M::U;
    const int sp = sl.cur[c] ^ prev;
    const size_t cb = (size_t)c * sy.Kmax + b;
    double D[RM * RM], JuL[RM * U], cp[RM];
    for (int i = 0; i < RM * RM; ++i) D[i] = w.Dw[cb * RM * RM + i];
    for (int i = 0; i < RM * U; ++i) JuL[i] = w.JuL[cb * RM * U + i];
    for (int i = 0; i < RM; ++i) cp[i] = w.cpad[cb * RM + i];
    int piv[RM];
    lu_factor<RM>(D, piv);
    lu_solve<RM, 1>(D, piv, cp);
    lu_solve<RM, U>(D, piv, JuL);
    for (int i = 0; i < RM; ++i) w.tpad[cb * RM + i] = cp[i];
    for (int i = 0; i < RM * U; ++i) w.Ew[cb * RM * U + i] = JuL[i];
    const double* jur = pick(sl.JuP, sp) + cb * RM * U;
    for (int a = 0; a < U; ++a) {
      double t2 = 0.0;
      for (int i = 0; i < RM; ++i) t2 += jur[i * U + a] * cp[i];
      w.sb[cb * U + a] = t2;
      for (int d = 0; d < U; ++d) {
        double t = 0.0;
        for (int i = 0; i < RM; ++i) t += jur[i * U + a] * JuL[i * U + d];
        w.Cb[(cb * U + a) * U + d] = t;
      }
    }
  }
};
template <class M, int RM>
struct KStateFactor {  // Cholesky of D = Jv Jv^T + diag, D^-1 dc/du, C_b, log det share  (:668-686)
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_FI CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    if (!w.ok[c]) return;
    (void)bd;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    const size_t cb = (size_t)c * sy.Kmax + b;
    double D[RM * RM], Ju[RM * U], E[RM * U];
    for (int i = 0; i < RM * RM; ++i) D[i] = w.Dw[cb * RM * RM + i];
    for (int i = 0; i < RM * U; ++i) Ju[i] = E[i] = pick(sl.JuP, s)[cb * RM * U + i];
    pick(sl.ldb, s)[cb] = chol_lower<RM>(D);
    for (int i = 0; i < RM * RM; ++i) pick(sl.facD, s)[cb * RM * RM + i] = D[i];
    cho_solve<RM, U>(D, E);
    for (int i = 0; i < RM * U; ++i) pick(sl.E, s)[cb * RM * U + i] = E[i];
    for (int a = 0; a < U; ++a)
      for (int d = 0; d < U; ++d) {
        double t = 0.0;
        for (int i = 0; i < RM; ++i) t += Ju[i * U + a] * E[i * U + d];
        w.Cb[(cb * U + a) * U + d] = t;
      }
  }
};

// Symmetric (Cholesky) block solve of an already stored padded vector: t = D^-1 v, s = Ju^T t
// (first half of lmult_by_inv_gram :920-930).  Input vector: work.cpad.
template <class M, int RM>
struct KSymBlk {
  Sys sy;
  Slots sl;
  Work w;
  int which, use_nw;
  CHMC_FI CHMC_HD void operator()(int tid) const {
    CHMC_CB_DECODE
    int which = this->which, qsel_unused = 0;
    if (use_nw ? !newton_select(w, c, which, qsel_unused) : !w.ok[c]) return;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    const size_t cb = (size_t)c * sy.Kmax + b;
    double Dl[RM * RM], t[RM];
    const double* fd = pick(sl.facD, s) + cb * RM * RM;
    for (int i = 0; i < RM * RM; ++i) Dl[i] = fd[i];
    for (int i = 0; i < RM; ++i) t[i] = w.cpad[cb * RM + i];
    cho_solve<RM, 1>(Dl, t);
    for (int i = 0; i < RM; ++i) w.tpad[cb * RM + i] = t[i];
    const double* ju = pick(sl.JuP, s) + cb * RM * U;
    for (int a = 0; a < U; ++a) {
      double acc = 0.0;
      for (int i = 0; i < RM; ++i) acc += ju[i * U + a] * t[i];
      w.sb[cb * U + a] = acc;
    }
  }
};

// Chain part of a Woodbury solve + the u-columns of the update.
//  SYM 0 (Newton :944-981):  C = I + sum Cb (LU), y = C^-1 sum s_b, lambda_b = t_b - Ew_b y
//  SYM 1 (Gram   :915-942):  C from the slot's Cholesky factor,     lambda_b = t_b - E_b y
// (lambda_b = D_b^-1 (v_b - Ju_b y) computed as D_b^-1 v_b - (D_b^-1 Ju_b) y.)
//  TGT 0: Newton / quasi-Newton update of the iterate: q_u -= d, mu_u += d, err = |c|_inf, ndq = |d_u|_inf
//  TGT 1: momentum projection: p_u -= d  (target p selected by psel: 0 slot `which` p, 1 work.pb)
//  TGT 2: no update (per-op API: multipliers only)
template <class M, int RM, int SYM, int TGT>
struct KSolveChain {
  Sys sy;
  Slots sl;
  Work w;
  int which, qsel, psel;  // which: slot holding the (previous-point) factors and dc/du
  CHMC_HD void operator()(int c) const {
    int which = this->which, qsel = this->qsel;
    if (TGT == 0 ? !newton_select(w, c, which, qsel) : !w.ok[c]) return;
    constexpr int U = M::U;
    const int s = sl.cur[c] ^ which;
    double sacc[U];
    for (int a = 0; a < U; ++a) sacc[a] = 0.0;
    for (int b = 0; b < sy.K; ++b)
      for (int a = 0; a < U; ++a) sacc[a] += w.sb[((size_t)c * sy.Kmax + b) * U + a];
    if (SYM) {
      double Lc[U * U];
      for (int i = 0; i < U * U; ++i) Lc[i] = pick(sl.facC, s)[(size_t)c * U * U + i];
      cho_solve<U, 1>(Lc, sacc);
    } else {
      double Cm[U * U];
      int piv[U];
      for (int i = 0; i < U * U; ++i) Cm[i] = 0.0;
      for (int i = 0; i < U; ++i) Cm[i * U + i] = 1.0;
      if (sy.m0)
        for (int i = 0; i < U * U; ++i) Cm[i] = sy.m0[i];
      for (int b = 0; b < sy.K; ++b)
        for (int i = 0; i < U * U; ++i) Cm[i] += w.Cb[((size_t)c * sy.Kmax + b) * U * U + i];
      lu_factor<U>(Cm, piv);
      lu_solve<U, 1>(Cm, piv, sacc);
    }
    double du[U];
    for (int a = 0; a < U; ++a) du[a] = 0.0;
    unsigned long long eb = 0ULL;
    for (int b = 0; b < sy.K; ++b) {
      const size_t cb = (size_t)c * sy.Kmax + b;
      const double* E = (SYM ? pick(sl.E, s) : w.Ew) + cb * RM * U;
      const double* ju = pick(sl.JuP, s) + cb * RM * U;
      for (int i = 0; i < RM; ++i) {
        double l = w.tpad[cb * RM + i];
        for (int a = 0; a < U; ++a) l -= E[i * U + a] * sacc[a];
        w.lampad[cb * RM + i] = l;
        for (int a = 0; a < U; ++a) du[a] += ju[i * U + a] * l;
        if (TGT == 0) {
          unsigned long long vb = absbits(w.cpad[cb * RM + i]);
          if (vb > eb) eb = vb;
        }
      }
    }
    if (TGT == 0) {
      double* q = (qsel ? w.qb : pick(sl.q, s ^ 1)) + (size_t)c * sy.Q;
      unsigned long long nb = 0ULL;
      for (int a = 0; a < U; ++a) {
        const double dq = metric_inv_u(sy, du, a);  // delta_q = metric.inv @ delta_mu (:1033-1041, :1105-1113)
        q[a] -= dq;
        unsigned long long vb = absbits(dq);
        if (vb > nb) nb = vb;
      }
      w.err[c] = bitsd(eb);
      w.ndq[c] = nb;
    } else if (TGT == 1) {
      double* p = (psel == 0 ? pick(sl.p, s) : psel == 1 ? w.pb : psel == 3 ? pick(sl.pg, s) : pick(sl.p, s ^ 1)) + (size_t)c * sy.Q;
      for (int a = 0; a < U; ++a) p[a] -= du[a];
    }
  }
};

// Column-parallel part of J^T lambda (rmult_by_jacob_constr :879-913) fused with its consumer; a "column max"
// kernel: work item (chain c, column group), the returned bit pattern is max-reduced per chain into *red(c)
// by the launcher (wave shuffle + LDS, one atomic per workgroup -- 2e7 same-address atomics serialise).
//  TGT 0: q -= d, ndq = max |d|              TGT 1: p -= d     TGT 2: out = d (work.pb, all columns incl. u)
//  VEC 2: one work item handles two adjacent columns with 16-byte accesses (needs even Q, NV, U, V0, V).
struct double2_ {
  double x, y;
};
// streaming 16-byte load (Jacobian rows are read once per sweep: keep them out of the caches' way)
CHMC_HD inline double2_ ld2_stream(const double* p) {
  double2_ r;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CHMC_PLAIN_STREAM)
  r.x = __builtin_nontemporal_load(p);
  r.y = __builtin_nontemporal_load(p + 1);
#else
  r.x = p[0], r.y = p[1];
#endif
  return r;
}
// Two adjacent components of a [B][Q] vector for the "row" launches (grid: column pairs x chains).  `wide`: the pair is
// 16-byte aligned (even Q) and complete; otherwise the components are accessed one by one (`two`: the second exists).
// Streaming hints as for the Jacobian rows: every vector of the integrator is far larger than the caches' share.
CHMC_HD inline double2_ ldv2(const double* p, bool wide, bool two) {
  double2_ r;
  if (wide) return ld2_stream(p);
  r.x = p[0], r.y = two ? p[1] : 0.0;
  return r;
}
CHMC_HD inline void stv2(double* p, double2_ v, bool wide, bool two) {
  if (wide) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
#else
    p[0] = v.x, p[1] = v.y;
#endif
    return;
  }
  p[0] = v.x;
  if (two) p[1] = v.y;
}
template <int RM, int TGT, int VEC>
struct KUpdate {
  static constexpr bool kFinish = false;
  Sys sy;
  Slots sl;
  Work w;
  int which, qsel, psel;
  CHMC_HD bool active(int c) const {
    int p_ = 0, q_ = 0;
    return TGT == 0 ? newton_select(w, c, p_, q_) : w.ok[c] != 0;
  }
  CHMC_HD double ncol_part2(int c, int col, int which) const {  // same with the first vector's multipliers (TGT 3)
    const int t = col - sy.NV;
    const int b = sy.obs2blk[t];
    const int j = t - sy.blk[b].obs0;
    return j < sy.blk[b].ny ? sigma_at(sy, pick(sl.q, sl.cur[c] ^ which) + (size_t)c * sy.Q) * w.lampad2[((size_t)c * sy.Kmax + b) * RM + j] : 0.0;
  }
  CHMC_HD unsigned long long* red(int c) const { return TGT == 0 ? &w.ndq[c] : nullptr; }
  CHMC_HD double ncol_part(int c, int col, int which) const {  // observation-noise columns: dc/dn = sigma on y rows (:601-608)
    const int t = col - sy.NV;
    const int b = sy.obs2blk[t];
    const int j = t - sy.blk[b].obs0;
    return j < sy.blk[b].ny ? sigma_at(sy, pick(sl.q, sl.cur[c] ^ which) + (size_t)c * sy.Q) * w.lampad[((size_t)c * sy.Kmax + b) * RM + j] : 0.0;
  }
  CHMC_HD unsigned long long operator()(int c, int idx) const {
    int which = this->which, qsel = this->qsel;
    if (TGT == 0) newton_select(w, c, which, qsel);  // 
    const int s = sl.cur[c] ^ which;
    const int col = idx * VEC;
    const size_t qi = (size_t)c * sy.Q + sy.U + col;
    double* tgt = TGT == 0 ? (qsel ? w.qb : pick(sl.q, s ^ 1))
                           : TGT == 1 ? (psel == 0 ? pick(sl.p, s) : psel == 1 ? w.pb : psel == 3 ? pick(sl.pg, s) : pick(sl.p, s ^ 1))
                           : TGT == 3 ? pick(sl.p, s) : w.pb;
    double d[VEC], old[VEC], d2[VEC], old2[VEC];  // (d2, old2: first vector of TGT 3)
    double* tgt2 = TGT == 3 ? pick(sl.p, s) : nullptr;
    if (TGT == 3) tgt = pick(sl.pg, s);  // TGT 3: p (multipliers lampad2) and pg (multipliers lampad) in one pass
    // issue the read-modify-write operands together with the Jacobian rows
    if (TGT != 2) {
      if (VEC == 2) {
        const double2_ o = *reinterpret_cast<const double2_*>(tgt + qi);
        old[0] = o.x, old[VEC - 1] = o.y;
      } else {
        old[0] = tgt[qi];
      }
    }
    if (TGT == 3) {
      if (VEC == 2) {
        const double2_ o = *reinterpret_cast<const double2_*>(tgt2 + qi);
        old2[0] = o.x, old2[VEC - 1] = o.y;
      } else {
        old2[0] = tgt2[qi];
      }
    }
    if (col < sy.NV) {
      const int g = col < sy.V0 ? 0 : (col - sy.V0) / sy.V / sy.S;
      const int b = sy.obs2blk[g];
      const double* lam = w.lampad + ((size_t)c * sy.Kmax + b) * RM;
      const double* lam2 = w.lampad2 + ((size_t)c * sy.Kmax + b) * RM;
      const double* Jv = pick(sl.Jv, s) + (size_t)c * RM * sy.NV + col;
      CHMC_UNROLL
      for (int k = 0; k < VEC; ++k) d[k] = 0.0, d2[k] = 0.0;
      // observation row i of a block is structurally zero in the intervals after its own observation, and slots
      // beyond the block's row count are padding: only rows [m, nrows) are read
      const int m = col < sy.V0 ? 0 : g - sy.blk[b].obs0;
      const int nr = sy.blk[b].nrows;
      CHMC_UNROLL
      for (int i = 0; i < RM; ++i) {
        if (i >= m && i < nr) {
          if (VEC == 2) {
            const double2_ jv = ld2_stream(Jv + (size_t)i * sy.NV);
            d[0] += jv.x * lam[i];
            d[VEC - 1] += jv.y * lam[i];
            if (TGT == 3) d2[0] += jv.x * lam2[i], d2[VEC - 1] += jv.y * lam2[i];
          } else {
            const double jv = Jv[(size_t)i * sy.NV];
            d[0] += jv * lam[i];
            if (TGT == 3) d2[0] += jv * lam2[i];
          }
        }
      }
    } else {
      CHMC_UNROLL
      for (int k = 0; k < VEC; ++k) {
        d[k] = ncol_part(c, col + k, which);
        if (TGT == 3) d2[k] = ncol_part2(c, col + k, which);
      }
    }
    unsigned long long r = 0ULL;
    CHMC_UNROLL
    for (int k = 0; k < VEC; ++k) {
      if (TGT == 0) {
        const unsigned long long vb = absbits(d[k]);
        r = vb > r ? vb : r;
      }
      old[k] = TGT == 2 ? d[k] : old[k] - d[k];
      if (TGT == 3) old2[k] -= d2[k];
    }
    if (VEC == 2) {
      double2_ o;
      o.x = old[0], o.y = old[VEC - 1];
      *reinterpret_cast<double2_*>(tgt + qi) = o;
      if (TGT == 3) {
        o.x = old2[0], o.y = old2[VEC - 1];
        *reinterpret_cast<double2_*>(tgt2 + qi) = o;
      }
    } else {
      tgt[qi] = old[0];
      if (TGT == 3) tgt2[qi] = old2[0];
    }
    return r;
  }
};

// The stored rows dc_i/dv_s = LF[m][i] . PB[s] of every chain's current state written out in full (Slots::Jv): only for
// the per-operator entry points of the ABI that hand rows to the caller or run the row-based passes; the stepping path
// works on the compact form and does not store the rows of the step columns at all.
template <int RM, int X, int V>
struct KRowsFromPB {
  Sys sy;
  Slots sl;
  CHMC_HD void operator()(int tid) const {
    const int TS = sy.T * sy.S;
    const int c = tid / TS, st = tid - c * TS;
    const int s = sl.cur[c];
    const int g = st / sy.S;
    const int b = sy.obs2blk[g];
    const int m = g - sy.blk[b].obs0;
    const double* lf = pick(sl.LF, s) + ((((size_t)c * sy.Kmax + b) * sy.NOBS + m) * RM) * X;
    const double* pb = pick(sl.PB, s) + ((size_t)c * TS + st) * (X * V);
    double* Jv = pick(sl.Jv, s) + (size_t)c * RM * sy.NV + sy.V0 + (size_t)st * V;
    double pbv[X * V];
    CHMC_UNROLL
    for (int k = 0; k < X * V; ++k) pbv[k] = pb[k];
    for (int i = 0; i < RM; ++i)
      CHMC_UNROLL
      for (int d = 0; d < V; ++d) {
        double t = 0.0;
        CHMC_UNROLL
        for (int a = 0; a < X; ++a) t += lf[i * X + a] * pbv[a * V + d];
        Jv[(size_t)i * sy.NV + d] = t;
      }
  }
};
// J^T lambda from the compact rows (Slots::PB, Slots::LF), the counterpart of KUpdate<RM, TGT, .>:
//   KMuF      muF[c][b][m] = sum_i lambda_i LF[m][i]   (TGT 3: muF2 from lampad2 as well; one work item per entry, tiny)
//   KUpdatePB target_v[s] -= muF[m(s)] . PB[s] for every step s; the v_0 and observation-noise columns as in KUpdate
//             (from the stored rows / sigma lambda).  Column-max launch over T S + V0 + (noisy ? T : 0) items.
//   TGT 0: Newton / quasi-Newton position update with max |delta| per chain;  TGT 1: a momentum-like vector (psel);
//   TGT 3: p (multipliers lampad2) and pg (multipliers lampad) in one pass.
template <int RM, int X, int TGT>
struct KMuF {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_FI CHMC_HD void operator()(int tid) const {
    const int a = tid % X;
    int r = tid / X;
    const int m = r % sy.NOBS;
    r /= sy.NOBS;
    const int b = r % sy.K, c = r / sy.K;
    int which = this->which, qsel_unused = 0;
    if (TGT == 0 ? !newton_select(w, c, which, qsel_unused) : !w.ok[c]) return;
    const size_t cb = (size_t)c * sy.Kmax + b;
    double t = 0.0, t2 = 0.0;
    if (m < sy.blk[b].nobs) {
      const double* lam = w.lampad + cb * RM;
      const double* lam2 = w.lampad2 + cb * RM;
      const double* lf = pick(sl.LF, sl.cur[c] ^ which) + (cb * sy.NOBS + m) * RM * X;
      const int nr = sy.blk[b].nrows;
      for (int i = 0; i < RM; ++i)
        if (i < nr) {
          t += lam[i] * lf[i * X + a];
          if (TGT == 3) t2 += lam2[i] * lf[i * X + a];
        }
    }
    w.muF[(cb * sy.NOBS + m) * X + a] = t;
    if (TGT == 3) w.muF2[(cb * sy.NOBS + m) * X + a] = t2;
  }
};
struct CheckArgs {  // KCheck's arguments riding on the Newton update pass (do_check == 0: no fused check)
  double ctol, ptol, dtol;
  int max_iters, do_check;
  int* iters_dst;  // the step's iteration counter of this retraction direction (KAddIters), or null
};
template <int RM, int X, int V, int TGT, int NS = 1>
struct KUpdatePB {  // NS: consecutive steps per work item (2 when S is even: both lie in the same observation interval)
  // TGT 0 with chk.do_check: the LAST workgroup to finish a chain's columns (ticket counter) runs KCheck for that chain --
  // the lax.while_loop condition needs max |delta q|, which is complete exactly then -- and adds the finished loop's
  // iteration count to the step's counter (KAddIters): two launches of every Newton round less.
  static constexpr bool kFinish = TGT == 0;
  Sys sy;
  Slots sl;
  Work w;
  int which, qsel, psel;
  CheckArgs chk;
  // TGT 3 with flow_rev: the reverse flow of the step's reversibility check (KFlow{1, 1, 0, -1}: work.qb = h2_flow(q, p, -dt))
  // for the columns of this pass, from the momentum it has just projected (the u-part: a KFlow launch over U columns;
  // standard splitting only)
  int flow_rev = 0;
  CHMC_HD bool has_finish() const { return TGT == 0 && chk.do_check != 0; }
  CHMC_HD void rev_flow2(int c, int s, size_t i, double px, double py) const {  // KFlow's expressions, a 16-byte pair
    const double2_ q0 = ld2_stream(pick(sl.q, s) + i);
    const double dt = -1.0 * w.dt[c];
    double2_ qn;
    qn.x = q0.x + dt * px, qn.y = q0.y + dt * py;
    stv2(w.qb + i, qn, true, true);
  }
  CHMC_HD void rev_flow1(int c, int s, size_t i, double px) const {  // ... one component
    const double dt = -1.0 * w.dt[c];
    w.qb[i] = pick(sl.q, s)[i] + dt * px;
  }
  CHMC_HD unsigned* ticket(int c) const { return w.ticket + c; }
  CHMC_HD void finish(int c, unsigned long long ndq_bits) const {
    w.ticket[c] = 0u;
    const int i = ++w.iters[c];
    const double err = w.err[c], ndq = bitsd(ndq_bits);
    const bool diverged = (err > chk.dtol) || (err != err);
    const bool converged = (err < chk.ctol) && (ndq < chk.ptol);
    if (i >= chk.max_iters || diverged || converged) {
      w.nw[c] = 0;
      const int st = converged ? 0 : (diverged ? 2 : 1);
      w.nstat[c] = st;
      if (st) w.ok[c] = 0, w.status[c] = st;
      if (chk.iters_dst) chk.iters_dst[c] += i;
    } else {
      atomic_add_i32(w.n_active, 1);
    }
  }
  CHMC_HD bool active(int c) const {
    int p_ = 0, q_ = 0;
    return TGT == 0 ? newton_select(w, c, p_, q_) : w.ok[c] != 0;
  }
  CHMC_HD unsigned long long* red(int c) const { return TGT == 0 ? &w.ndq[c] : nullptr; }
  CHMC_FI CHMC_HD unsigned long long operator()(int c, int idx) const {
    int which = this->which, qsel = this->qsel;
    if (TGT == 0) newton_select(w, c, which, qsel);  // 
    const int s = sl.cur[c] ^ which;
    const size_t off = (size_t)c * sy.Q + sy.U;
    double* tgt = (TGT == 0 ? (qsel ? w.qb : pick(sl.q, s ^ 1))
                            : TGT == 1 ? (psel == 0 ? pick(sl.p, s) : psel == 1 ? w.pb : psel == 3 ? pick(sl.pg, s) : pick(sl.p, s ^ 1))
                                       : pick(sl.pg, s)) + off;
    double* tgt2 = TGT == 3 ? pick(sl.p, s) + off : nullptr;  // TGT 3: p with the multipliers lampad2
    const int TS = sy.T * sy.S;
    unsigned long long r = 0ULL;
    if (idx < TS / NS) {
      const int st0 = idx * NS;
      const int g = st0 / sy.S;  // observation interval
      const int b = sy.obs2blk[g];
      const int m = g - sy.blk[b].obs0;
      const size_t mo = (((size_t)c * sy.Kmax + b) * sy.NOBS + m) * X;
      const double* pb = pick(sl.PB, s) + ((size_t)c * TS + st0) * (X * V);
      const size_t to = sy.V0 + (size_t)st0 * V;
      const bool wide = V == 2 && !((sy.Q | sy.U | sy.V0) & 1);
      double old[NS * V], old2[NS * V], pbv[NS * X * V], mu[X], mu2[X];
      // all loads first: the work item's bytes in flight hide the memory latency
      CHMC_UNROLL
      for (int e = 0; e < NS; ++e) {
        if (wide) {
          const double2_ o = *reinterpret_cast<const double2_*>(tgt + to + e * V);
          old[e * V] = o.x, old[e * V + V - 1] = o.y;
          if (TGT == 3) {
            const double2_ o2 = *reinterpret_cast<const double2_*>(tgt2 + to + e * V);
            old2[e * V] = o2.x, old2[e * V + V - 1] = o2.y;
          }
        } else {
          CHMC_UNROLL
          for (int k = 0; k < V; ++k) {
            old[e * V + k] = tgt[to + e * V + k];
            if (TGT == 3) old2[e * V + k] = tgt2[to + e * V + k];
          }
        }
      }
      if ((NS * X * V) % 2 == 0) {
        CHMC_UNROLL
        for (int k = 0; k < NS * X * V; k += 2) {
          const double2_ v = ld2_stream(pb + k);
          pbv[k] = v.x, pbv[k + 1 < NS * X * V ? k + 1 : k] = v.y;
        }
      } else {
        CHMC_UNROLL
        for (int k = 0; k < NS * X * V; ++k) pbv[k] = pb[k];
      }
      CHMC_UNROLL
      for (int a = 0; a < X; ++a) {
        mu[a] = w.muF[mo + a];
        if (TGT == 3) mu2[a] = w.muF2[mo + a];
      }
      CHMC_UNROLL
      for (int e = 0; e < NS; ++e)
        CHMC_UNROLL
        for (int k = 0; k < V; ++k) {
          double tt = 0.0, t2 = 0.0;
          CHMC_UNROLL
          for (int a = 0; a < X; ++a) {
            tt += mu[a] * pbv[e * X * V + a * V + k];
            if (TGT == 3) t2 += mu2[a] * pbv[e * X * V + a * V + k];
          }
          if (TGT == 0) {
            const unsigned long long vb = absbits(tt);
            r = vb > r ? vb : r;
          }
          old[e * V + k] -= tt;
          if (TGT == 3) old2[e * V + k] -= t2;
        }
      CHMC_UNROLL
      for (int e = 0; e < NS; ++e) {
        if (wide) {
          double2_ o;
          o.x = old[e * V], o.y = old[e * V + V - 1];
          *reinterpret_cast<double2_*>(tgt + to + e * V) = o;
          if (TGT == 3) {
            o.x = old2[e * V], o.y = old2[e * V + V - 1];
            *reinterpret_cast<double2_*>(tgt2 + to + e * V) = o;
            if (flow_rev) rev_flow2(c, s, off + to + e * V, o.x, o.y);
          }
        } else {
          CHMC_UNROLL
          for (int k = 0; k < V; ++k) {
            tgt[to + e * V + k] = old[e * V + k];
            if (TGT == 3) {
              tgt2[to + e * V + k] = old2[e * V + k];
              if (flow_rev) rev_flow1(c, s, off + to + e * V + k, old2[e * V + k]);
            }
          }
        }
      }
    } else {
      const int e = idx - TS / NS;
      double d = 0.0, d2 = 0.0;
      int col;
      if (e < sy.V0) {  // v_0 columns: the first block's stored rows
        col = e;
        const int b = sy.obs2blk[0];
        const double* lam = w.lampad + ((size_t)c * sy.Kmax + b) * RM;
        const double* lam2 = w.lampad2 + ((size_t)c * sy.Kmax + b) * RM;
        const double* Jv = pick(sl.Jv, s) + (size_t)c * RM * sy.NV + col;
        const int nr = sy.blk[b].nrows;
        for (int i = 0; i < RM; ++i)
          if (i < nr) {
            const double jv = Jv[(size_t)i * sy.NV];
            d += jv * lam[i];
            if (TGT == 3) d2 += jv * lam2[i];
          }
      } else {  // observation-noise columns: dc/dn = sigma on the y rows (:601-608)
        const int t = e - sy.V0;
        col = sy.NV + t;
        const int b = sy.obs2blk[t];
        const int j = t - sy.blk[b].obs0;
        if (j < sy.blk[b].ny) {
          const double sg = sigma_at(sy, pick(sl.q, s) + (size_t)c * sy.Q);
          d = sg * w.lampad[((size_t)c * sy.Kmax + b) * RM + j];
          if (TGT == 3) d2 = sg * w.lampad2[((size_t)c * sy.Kmax + b) * RM + j];
        }
      }
      if (TGT == 0) r = absbits(d);
      tgt[col] -= d;
      if (TGT == 3) {
        const double pn = tgt2[col] - d2;
        tgt2[col] = pn;
        if (flow_rev) rev_flow1(c, s, off + col, pn);
      }
    }
    return r;
  }
};

// ------------------------------------------------------------------------------------------
// find_initial_state_by_linear_interpolation (:1479-1547), batched: one work item per (chain, time step).  The path
// interpolates linearly between x_0 = generate_x_0(z, v_0) and the given states at the observation times; the
// one-step map is affine in v with a square, full-rank d forward_func / d v, so the noise increment of a step is
//   v_s = B(x_s)^-1 [ (x_{s+1} - x_s) - (forward_func(z, x_s, 0, delta) - x_s) ]          (solve_for_v_seq :1503-1526)
// Writes q = [u | v_0 | v_seq | n = 0] (:1533-1540) into state slot 0.
template <class M>
struct KInitInterp {
  Sys sy;
  double* q;           // [B][Q]
  const double* u;     // [B][U]
  const double* v0;    // [B][V0]
  const double* xobs;  // [B][T][X]  full states at the observation times (generate_x_obs_seq_init)
  CHMC_HD void operator()(int tid) const {
    constexpr int X = M::X, V = M::V;
    static_assert(X == V, "the interpolation solve needs a square noise matrix");
    const int TS = sy.T * sy.S;
    const int c = tid / TS, s = tid - c * TS;
    const int t = s / sy.S, i = s - t * sy.S;
    double* qc = q + (size_t)c * sy.Q;
    const double* uc = u + (size_t)c * sy.U;
    ChainConsts<M> cc;
    cc.init(uc, sy.dl);
    double xa[X], xs[X], dlt[X], f0[X], zero[V], A[X * X], Bm[X * V], rhs[X];
    if (t == 0) {
      M::gx0(cc.z, v0 + (size_t)c * sy.V0, xa);
    } else {
      for (int a = 0; a < X; ++a) xa[a] = xobs[((size_t)c * sy.T + t - 1) * X + a];
    }
    for (int a = 0; a < X; ++a) {
      dlt[a] = (xobs[((size_t)c * sy.T + t) * X + a] - xa[a]) / sy.S;
      xs[a] = xa[a] + i * dlt[a];
    }
    for (int a = 0; a < V; ++a) zero[a] = 0.0;
    M::step(cc.k, xs, zero, f0);
    M::jac_ab(cc.k, xs, zero, A, Bm);
    for (int a = 0; a < X; ++a) rhs[a] = dlt[a] - (f0[a] - xs[a]);
    int piv[X];
    lu_factor<X>(Bm, piv);
    lu_solve<X, 1>(Bm, piv, rhs);
    double* vq = qc + sy.U + sy.V0 + (size_t)s * V;
    for (int a = 0; a < V; ++a) vq[a] = rhs[a];
    if (s == 0) {
      for (int a = 0; a < sy.U; ++a) qc[a] = uc[a];
      for (int a = 0; a < sy.V0; ++a) qc[sy.U + a] = v0[(size_t)c * sy.V0 + a];
      if (sy.noisy)
        for (int a = 0; a < sy.T; ++a) qc[sy.U + sy.NV + a] = 0.0;
    }
  }
};

// ------------------------------------------------------------------------------------------
// Counter-based normal generator for the momentum refresh (IndependentMomentumTransition -> sample_momentum,
// :1256-1259): Philox4x32-10 (Salmon et al. 2011) keyed by the seed, counter = (component pair, draw index,
// global chain id), two 53-bit uniforms -> Box-Muller.  Stateless, so any sharding of the chains over GPUs and
// any launch geometry produce the same stream; tests/test_rng.py restates it in NumPy.
CHMC_HD inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                  uint32_t* out) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
struct KNormalFill {  // one work item per (chain, component pair); writes N(0, 1) into the state slot's momentum
  Sys sy;
  Slots sl;
  unsigned long long seed, draw;
  int chain_offset;
  CHMC_HD void operator()(int tid) const {
    const int npair = (sy.Q + 1) / 2;
    const int c = tid / npair, j = tid - c * npair;
    uint32_t r[4];
    philox4x32_10((uint32_t)j, (uint32_t)draw, (uint32_t)(c + chain_offset), (uint32_t)(draw >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const double u1 = (((uint64_t)r[0] << 21) ^ ((uint64_t)r[1] >> 11)) * (1.0 / 9007199254740992.0) +
                      (0.5 / 9007199254740992.0);  // (0, 1)
    const double u2 = (((uint64_t)r[2] << 21) ^ ((uint64_t)r[3] >> 11)) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    double* p = pick(sl.p, sl.cur[c]) + (size_t)c * sy.Q;
    p[2 * j] = rad * cos(ang);
    if (2 * j + 1 < sy.Q) p[2 * j + 1] = rad * sin(ang);
  }
};

// lax.while_loop condition (:1119-1127) evaluated per chain after each iteration, and the status mapping of
// the host wrappers (:1462-1476): 0 converged, 1 did not converge, 2 diverged / NaN.
struct KCheck {
  Work w;
  double ctol, ptol, dtol;
  int max_iters, B;
  int* iters_dst;  // the step's iteration counter of this retraction direction, or null: a chain's count is added when ITS
                   // loop ends, failed or not (KAddIters: "a chain that failed in this solve still reports its iterations")
  int slot;        // round & 3: the counter of this round
  CHMC_HD void operator()(int c) const {
    if (c == 0) w.n_active[(slot + 1) & 3] = 0;  // the next round's counter (its KCheck runs after this launch)
    if (!w.nw[c]) return;
    if (w.nw[c] != 1) {  // its time-parallel forward scan has not settled: not this round's iteration, but not finished
      atomic_add_i32(w.n_active + slot, 1);
      return;
    }
    const int i = ++w.iters[c];
    const double err = w.err[c], ndq = bitsd(w.ndq[c]);
    const bool diverged = (err > dtol) || (err != err);
    const bool converged = (err < ctol) && (ndq < ptol);
    if (i >= max_iters || diverged || converged) {
      w.nw[c] = 0;
      int st = converged ? 0 : (diverged ? 2 : 1);
      w.nstat[c] = st;
      if (st) w.ok[c] = 0, w.status[c] = st;
      if (iters_dst) iters_dst[c] += i;
    } else {
      atomic_add_i32(w.n_active + slot, 1);
    }
  }
};

struct KNewtonBegin {
  Work w;
  CHMC_HD void operator()(int c) const {
    w.nw[c] = w.ok[c];
    w.iters[c] = 0;
    w.err[c] = -1.0;
    w.ndq[c] = 0x7ff0000000000000ULL;  // +inf
    w.nstat[c] = 0;
    if (c == 0) w.n_active[0] = 0;  // the counter of round 0
  }
};

// After the last round the host enqueued: a chain that is STILL in the loop (the host stopped on a stale or wrong count, or
// ran out of rounds with a carried-over time-parallel scan) must not pass as converged -- KNewtonBegin left its projection
// status at 0.  It is reported as "did not converge" (:1393-1402) with the iterations it has done.
struct KNewtonEnd {
  Work w;
  int* iters_dst;
  CHMC_HD void operator()(int c) const {
    if (!w.nw[c]) return;
    w.nw[c] = 0;
    w.nstat[c] = 1;
    w.ok[c] = 0, w.status[c] = 1;
    if (iters_dst) iters_dst[c] += w.iters[c];
  }
};

// J w, one work item per (chain, block, row) (lmult_by_jacob_constr :822-877); input vector selected by
// vsel (0: slot p, 1: work.pb, 2: work.vin); result in work.cpad
template <int RM>
struct KJw {
  Sys sy;
  Slots sl;
  Work w;
  int which, vsel_;  // vsel_ | 256: the vector is multiplied by metric.inv first
  CHMC_HD void operator()(int tid) const {
    const int i = tid % RM;
    const int cbk = tid / RM;
    const int c = cbk / sy.K;
    const int b = cbk - c * sy.K;
    if (!w.ok[c]) return;
    const BlockDesc bd = sy.blk[b];
    const int s = sl.cur[c] ^ which;
    const bool minv = (vsel_ & 256) != 0;  // J (metric.inv @ vct) instead of J vct (:1243-1250)
    const int vsel = vsel_ & 255;
    const double* vct = (vsel == 0 ? pick(sl.p, s) : vsel == 1 ? w.pb : vsel == 2 ? w.vin : vsel == 4 ? pick(sl.pg, s) : pick(sl.p, s ^ 1)) + (size_t)c * sy.Q;
    const size_t cb = (size_t)c * sy.Kmax + b;
    double acc = 0.0;
    if (i < bd.nrows) {
      const double* ju = pick(sl.JuP, s) + (cb * RM + i) * sy.U;
      for (int a = 0; a < sy.U; ++a) acc += ju[a] * (minv ? metric_inv_u(sy, vct, a) : vct[a]);
      const double* Jv = pick(sl.Jv, s) + ((size_t)c * RM + i) * sy.NV + bd.col0;
      const double* wv = vct + sy.U + bd.col0;
      double a2 = 0.0;
      for (int k = 0; k < bd.ncols; ++k) a2 += Jv[k] * wv[k];
      acc += a2;
      if (sy.noisy && i < bd.ny) acc += sigma_at(sy, pick(sl.q, s) + (size_t)c * sy.Q) * vct[sy.U + sy.NV + bd.obs0 + i];
    }
    w.cpad[cb * RM + i] = acc;
  }
};

// element-wise pieces of the integrator (one work item per (chain, component))
// _step_a first half: p_out = p_in - h * dh1_dpos (:1192-1196), h = hfrac * dt[c]; q, grad and p_in are those
// of slot `which`; out_other != 0 writes the result into the other slot's p (leaving p_in untouched)
struct KKick {
  Sys sy;
  Slots sl;
  Work w;
  int which, out_other;
  double hfrac;
  CHMC_HD void operator()(int tid) const {
    const int c = tid / sy.Q;
    if (!w.ok[c]) return;
    const int s = sl.cur[c] ^ which;
    const double h = hfrac * w.dt[c];
    const double g = pick(sl.grad, s)[tid] + (sy.gaussian ? 0.0 : pick(sl.q, s)[tid]);
    const double pin = pick(sl.p, s)[tid];
    (out_other ? pick(sl.p, s ^ 1) : pick(sl.p, s))[tid] = pin - h * g;
  }
};
// pg = dh1_dpos of slot `which` (the kick direction of KKick), to be projected in place: P(q) dh1_dpos(q)
struct KInitPg {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_HD void operator()(int tid) const {
    const int c = tid / sy.Q;
    if (!w.ok[c]) return;
    const int s = sl.cur[c] ^ which;
    pick(sl.pg, s)[tid] = pick(sl.grad, s)[tid] + (sy.gaussian ? 0.0 : pick(sl.q, s)[tid]);
  }
};
// A(h dt) for a momentum that is already tangent at the slot's point: P (p - h dh1_dpos) = p - h P dh1_dpos
struct KKickPg {
  Sys sy;
  Slots sl;
  Work w;
  int which, out_other;
  double hfrac;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {  // row launch: components col, col + 1 of chain c
    const int s = sl.cur[c] ^ which;
    const double h = hfrac * w.dt[c];
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ p0 = ldv2(pick(sl.p, s) + i, wide, two), g = ldv2(pick(sl.pg, s) + i, wide, two);
    double2_ o;
    o.x = p0.x - h * g.x, o.y = p0.y - h * g.y;
    stv2((out_other ? pick(sl.p, s ^ 1) : pick(sl.p, s)) + i, o, wide, two);
  }
};
// h2_flow (:1222-1231) from slot `from` into (q_out, p_out): dst 0 = other slot, dst 1 = work (qb, pb)
struct KFlow {
  Sys sy;
  Slots sl;
  Work w;
  int from, dst, from_p_other;
  double sign;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {  // row launch: components col, col + 1 of chain c
    const int s = sl.cur[c] ^ from;
    const double dt = sign * w.dt[c];
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q0 = ldv2(pick(sl.q, s) + i, wide, two);
    const double2_ p0 = ldv2((from_p_other ? pick(sl.p, s ^ 1) : pick(sl.p, s)) + i, wide, two);
    double2_ qn, pn;
    if (sy.gaussian) {  // sin / cos of dt[c] are evaluated once per chain (KBegin), not per component
      const double sn = sign * w.sdt[c], cs = w.cdt[c];
      qn.x = q0.x * cs + sn * p0.x, qn.y = q0.y * cs + sn * p0.y;
      pn.x = p0.x * cs - sn * q0.x, pn.y = p0.y * cs - sn * q0.y;
    } else {
      qn.x = q0.x + dt * p0.x, qn.y = q0.y + dt * p0.y;
      pn = p0;
    }
    if (sy.m0 && col < sy.U) {  // pos += dt * metric.inv @ mom (:1208, :1231) on the u-part
      const double* pu = (from_p_other ? pick(sl.p, s ^ 1) : pick(sl.p, s)) + (size_t)c * sy.Q;
      qn.x = q0.x + dt * metric_inv_u(sy, pu, col);
      if (col + 1 < sy.U) qn.y = q0.y + dt * metric_inv_u(sy, pu, col + 1);
    }
    if (dst == 0) {
      stv2(pick(sl.q, s ^ 1) + i, qn, wide, two);
      // (standard splitting: the flow leaves the momentum as it is, and it already sits in the destination slot)
      if (sy.gaussian || !from_p_other) stv2(pick(sl.p, s ^ 1) + i, pn, wide, two);
    } else {
      stv2(w.qb + i, qn, wide, two);  // reverse-check flow: only the position is compared (KRevDiff), the momentum is not kept
    }
  }
};
// momentum correction after a successful projection: p -= dh2_flow_mom_dmom @ (mu / dt) (:1233-1238, :1465).
// The accumulated multiplier term mu = sum of the position updates is the distance the retraction moved the
// iterate, mu = h2_flow(q_prev, p) - q_new, so it is re-formed here from the two positions and the momentum instead
// of being carried (read + written) through every Newton iteration.
// ... fused with KInitPg of the new point in one pass (both only need the positions, the momentum and the gradient)
struct KMomFixInitPg {
  Sys sy;
  Slots sl;
  Work w;
  int which;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {  // row launch: components col, col + 1 of chain c
    const int s = sl.cur[c] ^ which;
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ qp = ldv2(pick(sl.q, s ^ 1) + i, wide, two), qn = ldv2(pick(sl.q, s) + i, wide, two);
    const double2_ pn = ldv2(pick(sl.p, s) + i, wide, two), gr = ldv2(pick(sl.grad, s) + i, wide, two);
    double sc;
    double2_ flow, po, go;
    if (sy.gaussian) {
      sc = w.cdt[c] / w.sdt[c];
      flow.x = (qp.x + w.sdt[c] * pn.x) / w.cdt[c], flow.y = (qp.y + w.sdt[c] * pn.y) / w.cdt[c];
    } else {
      sc = 1.0 / w.dt[c];
      flow.x = qp.x + w.dt[c] * pn.x, flow.y = qp.y + w.dt[c] * pn.y;
    }
    po.x = pn.x - sc * (flow.x - qn.x), po.y = pn.y - sc * (flow.y - qn.y);
    go.x = gr.x + (sy.gaussian ? 0.0 : qn.x), go.y = gr.y + (sy.gaussian ? 0.0 : qn.y);
    if (sy.m0 && col < sy.U) {
      // Block metric, u-part (U <= 8: chmc_set_metric): the multiplier term is mu_u = M_0 (flow - q_new)_u with
      // flow_u = q_prev_u + dt (M_0^-1 p)_u.  Every u-component needs all of p_u, which is updated in place, so the
      // first work item of the row handles the whole u-part and the others leave it alone.
      if (col != 0) {
        if (col + 1 >= sy.U && two) {  // odd dim_u: the pair's second component is the first one outside the u-part
          pick(sl.p, s)[i + 1] = po.y;
          pick(sl.pg, s)[i + 1] = go.y;
        }
        return;
      }
      const size_t cq = (size_t)c * sy.Q;
      const double* qpu = pick(sl.q, s ^ 1) + cq;
      const double* qnu = pick(sl.q, s) + cq;
      double* pu = pick(sl.p, s) + cq;
      double mv[8], pnew[8];
      for (int a = 0; a < sy.U; ++a) mv[a] = qpu[a] + w.dt[c] * metric_inv_u(sy, pu, a) - qnu[a];
      for (int a = 0; a < sy.U; ++a) pnew[a] = pu[a] - sc * metric_mul_u(sy, mv, a);
      for (int a = 0; a < sy.U; ++a) {
        pu[a] = pnew[a];
        pick(sl.pg, s)[cq + a] = pick(sl.grad, s)[cq + a] + qnu[a];
      }
      return;
    }
    stv2(pick(sl.p, s) + i, po, wide, two);
    stv2(pick(sl.pg, s) + i, go, wide, two);
  }
};
// KMomFixInitPg for the columns outside the step part only -- u, v_0 and the observation-noise components: work item `col` of
// the U + V0 + (Q - U - NV) edge columns -- when the step columns are corrected inside the J p pass (k_jw_pb<.., FIX>).
// (U, V0, NV even: a pair never straddles the gap.)
struct KMomFixEdges {
  KMomFixInitPg f;
  CHMC_HD bool active(int c) const { return f.active(c); }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {
    const int head = f.sy.U + f.sy.V0;
    f(c, col < head ? col : col - head + f.sy.U + f.sy.NV);
  }
};
// KKickPg into the other slot followed by KFlow from there, in one pass (tangent momentum at the start of a step)
struct KKickFlowPg {
  Sys sy;
  Slots sl;
  Work w;
  double hfrac;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {  // row launch: components col, col + 1 of chain c
    const int s = sl.cur[c];
    const double h = hfrac * w.dt[c];
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q0 = ldv2(pick(sl.q, s) + i, wide, two), pp = ldv2(pick(sl.p, s) + i, wide, two);
    const double2_ g = ldv2(pick(sl.pg, s) + i, wide, two);
    double2_ p0, qn, pn;
    p0.x = pp.x - h * g.x, p0.y = pp.y - h * g.y;
    if (sy.gaussian) {
      const double sn = w.sdt[c], cs = w.cdt[c];
      qn.x = q0.x * cs + sn * p0.x, qn.y = q0.y * cs + sn * p0.y;
      pn.x = p0.x * cs - sn * q0.x, pn.y = p0.y * cs - sn * q0.y;
    } else {
      qn.x = q0.x + w.dt[c] * p0.x, qn.y = q0.y + w.dt[c] * p0.y;
      pn = p0;
    }
    if (sy.m0 && col < sy.U) {  // u-part of the flow with the block metric: q_u += dt * M_0^-1 (p - h pg)_u
      const double* pu = pick(sl.p, s) + (size_t)c * sy.Q;
      const double* gu = pick(sl.pg, s) + (size_t)c * sy.Q;
      double t0 = 0.0, t1 = 0.0;
      for (int b = 0; b < sy.U; ++b) {
        const double pb = pu[b] - h * gu[b];
        t0 += sy.m0[sy.U * sy.U + col * sy.U + b] * pb;
        if (col + 1 < sy.U) t1 += sy.m0[sy.U * sy.U + (col + 1) * sy.U + b] * pb;
      }
      qn.x = q0.x + w.dt[c] * t0;
      if (col + 1 < sy.U) qn.y = q0.y + w.dt[c] * t1;
    }
    stv2(pick(sl.q, s ^ 1) + i, qn, wide, two);
    stv2(pick(sl.p, s ^ 1) + i, pn, wide, two);
  }
};
// The closing A(dt/2) of the previous step of a trajectory and KKickFlowPg of the next one in ONE pass (chmc_leapfrog_steps,
// lock-step path): p1 = p - h pg goes back to the state slot (the state a failing step leaves behind), the flow starts from
// p1 - h pg.  Same operations in the same order as KKickPg followed by KKickFlowPg; two vector passes less per step.
// (Not used with a block metric: its u-part reads all of p_u while other work items would be rewriting it.)
struct KKick2FlowPg {
  Sys sy;
  Slots sl;
  Work w;
  double hfrac;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_FI CHMC_HD void operator()(int c, int col) const {  // row launch: components col, col + 1 of chain c
    const int s = sl.cur[c];
    const double h = hfrac * w.dt[c];
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q0 = ldv2(pick(sl.q, s) + i, wide, two), pp = ldv2(pick(sl.p, s) + i, wide, two);
    const double2_ g = ldv2(pick(sl.pg, s) + i, wide, two);
    double2_ p1, p0, qn, pn;
    p1.x = pp.x - h * g.x, p1.y = pp.y - h * g.y;  // KKickPg of the step before
    p0.x = p1.x - h * g.x, p0.y = p1.y - h * g.y;
    if (sy.gaussian) {
      const double sn = w.sdt[c], cs = w.cdt[c];
      qn.x = q0.x * cs + sn * p0.x, qn.y = q0.y * cs + sn * p0.y;
      pn.x = p0.x * cs - sn * q0.x, pn.y = p0.y * cs - sn * q0.y;
    } else {
      qn.x = q0.x + w.dt[c] * p0.x, qn.y = q0.y + w.dt[c] * p0.y;
      pn = p0;
    }
    stv2(pick(sl.p, s) + i, p1, wide, two);
    stv2(pick(sl.q, s ^ 1) + i, qn, wide, two);
    stv2(pick(sl.p, s ^ 1) + i, pn, wide, two);
  }
};
// reverse check distance max |q_back - q_start| (mici maximum_norm); column-max kernel, two components per work item
struct KRevDiff {
  static constexpr bool kFinish = false;
  Sys sy;
  Slots sl;
  Work w;
  CHMC_HD bool active(int c) const { return w.ok[c] != 0; }
  CHMC_HD unsigned long long* red(int c) const { return &w.rev[c]; }
  CHMC_FI CHMC_HD unsigned long long operator()(int c, int idx) const {
    const int col = 2 * idx;
    const size_t i = (size_t)c * sy.Q + col;
    const double* qs = pick(sl.q, sl.cur[c]);
    if (col + 1 < sy.Q && (sy.Q & 1) == 0) {
      const double2_ a = *reinterpret_cast<const double2_*>(w.qb + i), b = *reinterpret_cast<const double2_*>(qs + i);
      const unsigned long long u = absbits(a.x - b.x), v = absbits(a.y - b.y);
      return u > v ? u : v;
    }
    unsigned long long r = absbits(w.qb[i] - qs[i]);
    if (col + 1 < sy.Q) {
      const unsigned long long v = absbits(w.qb[i + 1] - qs[i + 1]);
      r = v > r ? v : r;
    }
    return r;
  }
};
struct KRevCheck {
  Work w;
  double tol;
  CHMC_FI CHMC_HD void operator()(int c) const {
    if (!w.ok[c]) return;
    const double r = bitsd(w.rev[c]);
    if (!(r <= tol)) w.ok[c] = 0, w.status[c] = 3;
  }
};
struct KCommit {  // accept: the proposal slot becomes the state slot
  Slots sl;
  Work w;
  CHMC_FI CHMC_HD void operator()(int c) const {
    if (w.ok[c]) sl.cur[c] ^= 1;
  }
};
struct KBegin {
  Work w;
  const int* active;
  const double* dt;
  double scale;  // work.dt = scale * dt: the time step of one inner h2-flow step, dt / n_inner_step (mici _step_b)
  CHMC_HD void operator()(int c) const {
    w.ok[c] = active ? (active[c] != 0) : 1;
    w.status[c] = w.ok[c] ? 0 : -1;
    if (dt) {
      const double h = scale * dt[c];
      w.dt[c] = h;
      w.sdt[c] = sin(h);
      w.cdt[c] = cos(h);
    }
    w.rev[c] = 0ULL;
  }
};
// n_inner_step > 1: accumulate the iteration counts of the inner steps; between inner steps the new point becomes
// state_prev (mici _step_b: `state_prev = state.copy()`), i.e. the proposal slot becomes the state slot
struct KAddIters {
  Work w;
  int* dst;
  CHMC_HD void operator()(int c) const {
    if (w.ok[c] || w.status[c] > 0) dst[c] += w.iters[c];  // (a chain that failed in this solve still reports its iterations)
  }
};
struct KCommitInner {
  Slots sl;
  Work w;
  int* ncommit;
  CHMC_HD void operator()(int c) const {
    if (w.ok[c]) {
      sl.cur[c] ^= 1;
      ncommit[c] += 1;
      w.rev[c] = 0ULL;  // the reported reverse-check distance is that of the last inner step
    }
  }
};
struct KCopyPadMasked {  // dst[c][:] = src[c][:] for the chains of the view (block-padded [B][n] arrays)
  Work w;
  double* dst;
  const double* src;
  int n;
  CHMC_HD void operator()(int tid) const {
    if (w.ok[tid / n]) dst[tid] = src[tid];
  }
};

// One leaf of a dynamic (no-U-turn) trajectory tree, batched: everything the caller's tree bookkeeping needs from the
// state the integrator has just produced, in ONE pass over its position and momentum (see chmc_tree_leaf in
// include/chmc.h).  Row-sum launch: f(c, col, acc) handles components col, col + 1 of chain c and adds into the
// per-item accumulators.  The sub-tree spans that end at this (odd) leaf b are nested; span k starts at the leaf a_k
// recorded in checkpoint slot lo + k (slot lo: the largest span) and the span of slot lo + k + 1 is its right half.
// With v(.) = dh_dmom = metric.inv @ mom (:1204-1208) and S = the momentum sum of the sub-tree's leaves up to b:
//   acc[6k + 0], acc[6k + 1] = v(p_a) . rho, v(p_b) . rho,   rho = sum of the momenta of leaves a .. b
//                                                           (= S - ck_sum[a] + ck_p[a]);
// and, when `ck_end` is given (Mici's additional sub-tree checks across the two halves of a span of >= 4 leaves,
// m = last leaf of the left half, m + 1 = first leaf of the right half, both known from the checkpoints):
//   acc[6k + 2], acc[6k + 3] = v(p_a) . rho1, v(p_{m+1}) . rho1,   rho1 = (momenta of a .. m) + p_{m+1}
//   acc[6k + 4], acc[6k + 5] = v(p_m) . rho2, v(p_b) . rho2,       rho2 = (momenta of m+1 .. b) + p_m.
// After the checks the leaf's momentum is recorded in ck_end[lo]: b is the last leaf of the left half of the next
// larger span that starts at a_lo.
#define CHMC_TREE_MAXCHK 10
#define CHMC_ROWSUM_MAX (6 * CHMC_TREE_MAXCHK)
struct KTreeLeaf {
  Sys sy;
  Slots sl;
  const int* run;    // [B] chains that took this leaf
  const int* take;   // [B] chains whose sub-tree proposal becomes this leaf
  double* sub_prop_q;  // [B][Q]
  double* sub_sum;     // [B][Q] momentum sum of the sub-tree's leaves so far
  double* ck_p;        // [D][B][Q] momentum at the first leaf of a pending span
  double* ck_sum;      // [D][B][Q] sub_sum at that leaf
  double* ck_end;      // [D][B][Q] momentum at the last leaf of the left half of the pending span of a slot (or null)
  int store, lo, nchk;  // store: checkpoint slot this leaf is recorded in (-1: none); checks against slots lo .. lo + nchk - 1
  CHMC_HD bool active(int c) const { return run[c] != 0; }
  CHMC_HD double2_ vel(const double* base, size_t cq, int col, double2_ p) const {
    // metric.inv @ p: only the u-part of the block metric differs from p
    if (sy.m0 && col < sy.U) {
      p.x = metric_inv_u(sy, base + cq, col);
      if (col + 1 < sy.U) p.y = metric_inv_u(sy, base + cq, col + 1);
    }
    return p;
  }
  CHMC_HD void operator()(int c, int col, double* acc) const {
    const int s = sl.cur[c];
    const size_t cq = (size_t)c * sy.Q, i = cq + col, BQ = (size_t)sy.B * sy.Q;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q = ldv2(pick(sl.q, s) + i, wide, two), p = ldv2(pick(sl.p, s) + i, wide, two);
    double2_ S = ldv2(sub_sum + i, wide, two);
    S.x += p.x, S.y += p.y;
    stv2(sub_sum + i, S, wide, two);
    if (take[c]) stv2(sub_prop_q + i, q, wide, two);
    if (store >= 0) {
      stv2(ck_p + store * BQ + i, p, wide, two);
      stv2(ck_sum + store * BQ + i, S, wide, two);
    }
    const double2_ vp = vel(pick(sl.p, s), cq, col, p);
    double2_ a_r, cs_r;  // checkpoint of the right half (the next smaller span), carried from slot to slot
    a_r.x = a_r.y = cs_r.x = cs_r.y = 0.0;
    // from the smallest span (slot lo + nchk - 1) to the largest (slot lo)
    CHMC_UNROLL
    for (int kk = 0; kk < CHMC_TREE_MAXCHK; ++kk) {
      const int k = nchk - 1 - kk;
      if (k >= 0) {
        const double* cp = ck_p + (size_t)(lo + k) * BQ;
        const double2_ a = ldv2(cp + i, wide, two), cs = ldv2(ck_sum + (size_t)(lo + k) * BQ + i, wide, two);
        const double2_ va = vel(cp, cq, col, a);
        const double sx = S.x - cs.x + a.x, sy_ = two ? S.y - cs.y + a.y : 0.0;
        acc[6 * k] += va.x * sx + (two ? va.y * sy_ : 0.0);
        acc[6 * k + 1] += vp.x * sx + (two ? vp.y * sy_ : 0.0);
        if (ck_end && kk > 0) {
          const double* ce = ck_end + (size_t)(lo + k) * BQ;
          const double* cr = ck_p + (size_t)(lo + k + 1) * BQ;
          const double2_ pm = ldv2(ce + i, wide, two);
          const double2_ vm = vel(ce, cq, col, pm), vr = vel(cr, cq, col, a_r);
          const double r1x = cs_r.x - cs.x + a.x, r1y = two ? cs_r.y - cs.y + a.y : 0.0;
          const double r2x = S.x - cs_r.x + a_r.x + pm.x, r2y = two ? S.y - cs_r.y + a_r.y + pm.y : 0.0;
          acc[6 * k + 2] += va.x * r1x + (two ? va.y * r1y : 0.0);
          acc[6 * k + 3] += vr.x * r1x + (two ? vr.y * r1y : 0.0);
          acc[6 * k + 4] += vm.x * r2x + (two ? vm.y * r2y : 0.0);
          acc[6 * k + 5] += vp.x * r2x + (two ? vp.y * r2y : 0.0);
        }
        a_r = a, cs_r = cs;
      }
    }
    // (the u-part of ck_end[lo] is read by every work item of the chain's first columns through the block metric: it is
    // recorded by the second stage, KTreeLeafFinish, after this pass)
    if (ck_end && nchk > 0) {
      double* ce = ck_end + (size_t)lo * BQ + i;
      if (col >= sy.U) stv2(ce, p, wide, two);
      else if (two && col + 1 >= sy.U) ce[1] = p.y;
    }
  }
};
// second stage: the workgroup partials of a chain are added in a fixed order (reproducible sums);
// out [B][2 nchk] = the two criterion values per checkpoint
struct KTreeLeafFinish {
  Sys sy;
  Slots sl;
  const int* run;
  const double* partial;  // [B][nwg][nacc]
  int nwg, nacc;
  double* out;            // [B][nacc]
  double* ck_end;         // u-part of ck_end[lo] <- momentum (see KTreeLeaf), or null
  int lo;
  CHMC_HD void operator()(int c) const {
    double* o = out + (size_t)c * nacc;
    for (int a = 0; a < nacc; ++a) {
      double t = 0.0;
      if (run[c])
        for (int g = 0; g < nwg; ++g) t += partial[((size_t)c * nwg + g) * nacc + a];
      o[a] = t;
    }
    if (ck_end && run[c]) {
      const double* p = pick(sl.p, sl.cur[c]) + (size_t)c * sy.Q;
      double* ce = ck_end + ((size_t)lo * sy.B + c) * sy.Q;
      for (int a = 0; a < sy.U; ++a) ce[a] = p[a];
    }
  }
};
// ---- per-chain decisions of the batched dynamic transition on the device (chmc_tree_begin / _subtree / _step in
// include/chmc.h; the logic of mici's MultinomialDynamicIntegrationTransition._build_tree base case, per chain)
struct TreeState {
  double *h0, *sub_logw, *sum_acc, *u;  // [B] Hamiltonian at the tree's root, log weight of the current sub-tree, sum of
                                        // the leaves' acceptance probabilities, this leaf's uniform draw
  int *alive, *run, *take, *n_step, *failed, *diverged;  // [B]
  int* n_running;                                        // [1]
  // per-doubling state (chmc_tree_doubling_begin / _end): log weight of the whole tree, direction of the current
  // doubling, which tree edge the context's chain state sits on, whether the proposal has moved, doublings completed,
  // chains whose current sub-tree completed / whose proposal moves to it / whose state has to be brought to the other edge
  double *logw, *u2;
  int *fwd, *at_pos, *at_neg, *moved, *depth, *done, *acc, *need;
  int* counts;  // [2] chains alive, chains that need an edge switch
};
CHMC_HD inline double log_add_exp(double a, double b) {  // log(exp(a) + exp(b)), the formula of numpy.logaddexp
  if (a == b) return a + 0.6931471805599453;             // (also covers -inf, -inf)
  const double d = a - b;
  if (d > 0.0) return a + log1p(exp(-d));
  if (d <= 0.0) return b + log1p(exp(d));
  return a + b;  // NaN
}
struct KTreeBegin {
  TreeState t;
  const double* ham;    // [B][3] chmc_hamiltonian of the current states
  CHMC_HD void operator()(int c) const {
    const double h = ham[(size_t)c * 3];
    t.h0[c] = h;
    t.alive[c] = (h - h == 0.0) ? 1 : 0;  // finite
    t.run[c] = 0, t.take[c] = 0, t.n_step[c] = 0, t.failed[c] = 0, t.diverged[c] = 0;
    t.sum_acc[c] = 0.0;
    t.sub_logw[c] = -__builtin_huge_val();
    t.logw[c] = -h, t.at_pos[c] = 1, t.at_neg[c] = 1, t.moved[c] = 0, t.depth[c] = 0, t.done[c] = 0, t.acc[c] = 0, t.need[c] = 0;
  }
};
// ---- the per-doubling logic of the transition on the device (mici _build_tree's callers: direction draw, biased
// progressive sampling between the old tree and the new sub-tree, whole-tree no-U-turn criterion; dynamic.py held these
// on the host in round 2)
struct KTreeDoubleBegin {  // direction of this doubling, edge switch needed?, new sub-tree (KTreeSubBegin)
  TreeState t;
  CHMC_HD void operator()(int c) const {
    t.run[c] = t.alive[c];
    t.sub_logw[c] = -__builtin_huge_val();
    t.done[c] = 0, t.acc[c] = 0;
    int need = 0;
    if (t.alive[c]) {
      const int fwd = t.u2[c] < 0.5 ? 1 : 0;
      need = fwd ? !t.at_pos[c] : !t.at_neg[c];
      t.fwd[c] = fwd, t.at_pos[c] = fwd, t.at_neg[c] = !fwd;
      atomic_add_i32(t.counts, 1);
      if (need) atomic_add_i32(t.counts + 1, 1);
    }
    t.need[c] = need;
  }
};
struct KTreeRestoreEdge {  // chain state <- the tree edge the doubling extends from, for the chains that are on the other one
  Sys sy;
  Slots sl;
  TreeState t;
  const double *neg_q, *neg_p, *pos_q, *pos_p;
  CHMC_HD void operator()(int tid) const {
    const int c = tid / sy.Q;
    if (!t.need[c]) return;
    const int s = sl.cur[c];
    pick(sl.q, s)[tid] = (t.fwd[c] ? pos_q : neg_q)[tid];
    pick(sl.p, s)[tid] = (t.fwd[c] ? pos_p : neg_p)[tid];
  }
};
struct KTreeDoubleDecide {  // after the sub-tree's leaves: biased progressive sampling (old tree vs new sub-tree)
  TreeState t;
  int depth;
  CHMC_HD void operator()(int c) const {
    const int done = t.run[c] != 0;  // the sub-tree completed without terminating
    t.done[c] = done, t.acc[c] = 0;
    if (!done) return;
    const double d = t.sub_logw[c] - t.logw[c];
    const int acc = t.u2[c] < exp(d < 0.0 ? d : 0.0) ? 1 : 0;
    t.acc[c] = acc;
    if (acc) t.moved[c] = 1;
    t.logw[c] = log_add_exp(t.logw[c], t.sub_logw[c]);
    t.depth[c] = depth + 1;
  }
};
// one pass over the state of the chains whose sub-tree completed: proposal, momentum sum, the tree's new edge, and the two
// sides of the whole-tree criterion dh_dmom(edge) . sum of momenta (row-sum launch: acc[0] negative edge, acc[1] positive)
struct KTreeDoubleRows {
  Sys sy;
  Slots sl;
  TreeState t;
  double *prop_q, *sum_mom, *neg_q, *neg_p, *pos_q, *pos_p;
  const double *sub_prop_q, *sub_sum;
  CHMC_HD bool active(int c) const { return t.done[c] != 0; }
  CHMC_HD double2_ vel(const double* base, size_t cq, int col, double2_ p) const {
    if (sy.m0 && col < sy.U) {
      p.x = metric_inv_u(sy, base + cq, col);
      if (col + 1 < sy.U) p.y = metric_inv_u(sy, base + cq, col + 1);
    }
    return p;
  }
  CHMC_HD void operator()(int c, int col, double* acc) const {
    const int s = sl.cur[c], fwd = t.fwd[c];
    const size_t cq = (size_t)c * sy.Q, i = cq + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q = ldv2(pick(sl.q, s) + i, wide, two), p = ldv2(pick(sl.p, s) + i, wide, two);
    if (t.acc[c]) stv2(prop_q + i, ldv2(sub_prop_q + i, wide, two), wide, two);
    double2_ S = ldv2(sum_mom + i, wide, two);
    const double2_ ss = ldv2(sub_sum + i, wide, two);
    S.x += ss.x, S.y += ss.y;
    stv2(sum_mom + i, S, wide, two);
    // the state after the sub-tree's last leaf is the tree's new edge on the side the doubling went; the other edge
    // is the stored one (its u-part is read through the block metric: the stored buffer of the OTHER side is not written here)
    const double* nbase = fwd ? neg_p : pick(sl.p, s);
    const double* pbase = fwd ? pick(sl.p, s) : pos_p;
    const double2_ pn = fwd ? ldv2(neg_p + i, wide, two) : p;
    const double2_ pp = fwd ? p : ldv2(pos_p + i, wide, two);
    stv2((fwd ? pos_q : neg_q) + i, q, wide, two);
    stv2((fwd ? pos_p : neg_p) + i, p, wide, two);
    const double2_ vn = vel(nbase, cq, col, pn), vp = vel(pbase, cq, col, pp);
    acc[0] += vn.x * S.x + (two ? vn.y * S.y : 0.0);
    acc[1] += vp.x * S.x + (two ? vp.y * S.y : 0.0);
  }
};
struct KTreeDoubleFinish {  // whole-tree no-U-turn criterion (riemannian_no_u_turn_criterion) and the count of live chains
  TreeState t;
  const double* partial;  // [B][nwg][2]
  int nwg;
  CHMC_HD void operator()(int c) const {
    if (t.done[c]) {
      double d0 = 0.0, d1 = 0.0;
      for (int g = 0; g < nwg; ++g) d0 += partial[((size_t)c * nwg + g) * 2], d1 += partial[((size_t)c * nwg + g) * 2 + 1];
      if (d0 < 0.0 || d1 < 0.0) t.alive[c] = 0;
    }
    if (t.alive[c]) atomic_add_i32(t.counts, 1);
  }
};
struct KTreeSubBegin {  // a new sub-tree: every live chain runs, empty multinomial weight
  TreeState t;
  CHMC_HD void operator()(int c) const {
    t.run[c] = t.alive[c];
    t.sub_logw[c] = -__builtin_huge_val();
  }
};
// after the integrator step of a leaf: integrator errors and divergence (delta_h > max_delta_h, NaN counts as
// divergent) end the chain's tree; otherwise the leaf joins the sub-tree (step count, acceptance statistic, multinomial
// weight) and becomes its proposal with probability exp(-h) / (sub-tree weight including this leaf)
struct KTreeDecide {
  TreeState t;
  const int* status;  // of the step just taken
  const double* ham;  // [B][3] at the new states
  double max_delta_h;
  CHMC_HD void operator()(int c) const {
    t.take[c] = 0;
    if (!t.run[c]) return;
    const double h = ham[(size_t)c * 3], h0 = t.h0[c];
    const bool bad = status[c] != 0;
    const bool div = !bad && !((h - h0) <= max_delta_h);
    if (bad) t.failed[c] = 1;
    if (div) t.diverged[c] = 1;
    if (bad || div) {
      t.alive[c] = 0, t.run[c] = 0;
      return;
    }
    t.n_step[c] += 1;
    const double d = h0 - h;
    t.sum_acc[c] += fmin(1.0, exp(d < 0.0 ? d : 0.0));
    const double nw = log_add_exp(t.sub_logw[c], -h);
    t.take[c] = t.u[c] < exp(-h - nw) ? 1 : 0;
    t.sub_logw[c] = nw;
  }
};
// after KTreeLeaf / KTreeLeafFinish of an odd leaf: a negative criterion value on any checked span ends the chain's tree
struct KTreeTurn {
  TreeState t;
  const double* out;  // [B][nacc]
  int nacc;
  CHMC_HD void operator()(int c) const {
    if (!t.run[c]) return;
    bool turn = false;
    for (int a = 0; a < nacc; ++a) turn = turn || out[(size_t)c * nacc + a] < 0.0;
    if (turn) t.alive[c] = 0, t.run[c] = 0;
  }
};
struct KTreeCount {
  TreeState t;
  int B;
  CHMC_HD void operator()(int tid) const {
    if (tid != 0) return;
    int n = 0;
    for (int c = 0; c < B; ++c) n += t.run[c] != 0;
    t.n_running[0] = n;
  }
};
// q.q and p.p of the current state for the Hamiltonian (:1186-1202), row-sum launch: acc[0] += q.q, acc[1] += p.p
struct KNormRow {
  Sys sy;
  Slots sl;
  CHMC_HD bool active(int) const { return true; }
  CHMC_HD void operator()(int c, int col, double* acc) const {
    const int s = sl.cur[c];
    const size_t i = (size_t)c * sy.Q + col;
    const bool two = col + 1 < sy.Q, wide = two && !(sy.Q & 1);
    const double2_ q = ldv2(pick(sl.q, s) + i, wide, two), p = ldv2(pick(sl.p, s) + i, wide, two);
    acc[0] += q.x * q.x + (two ? q.y * q.y : 0.0);
    acc[1] += p.x * p.x + (two ? p.y * p.y : 0.0);
  }
};

// sample_momentum with the block metric: mom = metric.sqrt @ n (:1257), i.e. the u-part becomes chol(M_0) n_u
struct KMetricSqrtU {
  Sys sy;
  Slots sl;
  CHMC_HD void operator()(int c) const {
    double* pu = pick(sl.p, sl.cur[c]) + (size_t)c * sy.Q;
    const double* L = sy.m0 + 2 * sy.U * sy.U;
    for (int a = sy.U - 1; a >= 0; --a) {  // lower triangular: row a only needs entries <= a, so go bottom-up in place
      double t = 0.0;
      for (int b = 0; b <= a; ++b) t += L[a * sy.U + b] * pu[b];
      pu[a] = t;
    }
  }
};
struct KHamiltonian {
  Sys sy;
  Slots sl;
  Work w;
  int npart;
  double* out;  // [B][3]: h, 1/2 q.q, 1/2 p.p
  CHMC_HD void operator()(int c) const {
    double qq = 0.0, pp = 0.0;
    for (int j = 0; j < npart; ++j) qq += w.part[((size_t)c * npart + j) * 2], pp += w.part[((size_t)c * npart + j) * 2 + 1];
    const double ld = sl.logdet[sl.cur[c]][c];
    if (sy.m0) {  // h2 = mom @ metric.inv @ mom / 2 (:1202): the u-part's quadratic form replaces its plain square
      const double* pu = pick(sl.p, sl.cur[c]) + (size_t)c * sy.Q;
      for (int a = 0; a < sy.U; ++a) pp += pu[a] * (metric_inv_u(sy, pu, a) - pu[a]);
    }
    out[c * 3 + 1] = 0.5 * qq;
    out[c * 3 + 2] = 0.5 * pp;
    out[c * 3] = 0.5 * qq + ld + 0.5 * pp;  // h1 + h2 is the same sum for both splittings
  }
};


// Adam-based initial states (find_initial_state_by_gradient_descent_noisy_system, sde/mici_extensions.py:1679-1801), the
// device-resident loop: what the restart rules need of a chain's parameters and gradient (row sums), and the Adam step
// (jax.example_libraries.optimizers.adam: m <- b1 m + (1 - b1) g, v <- b2 v + (1 - b2) g^2, x <- x - lr m_hat / (sqrt(v_hat) + eps))
// on the caller's device buffers.  A non-finite gradient entry enters the moments as 0 (its chain is restarted by the host).
struct KAdamRow {
  const double* uv;
  const double* g;
  int n;
  CHMC_HD bool active(int) const { return true; }
  CHMC_HD void operator()(int c, int col, double* acc) const {
    const size_t i = (size_t)c * n + col;
    const bool two = col + 1 < n;
    const double u0 = uv[i], u1 = two ? uv[i + 1] : 0.0, g0 = g[i], g1 = two ? g[i + 1] : 0.0;
    acc[0] += u0 * u0 + u1 * u1;
    acc[1] += (g0 - g0 == 0.0 ? 0.0 : 1.0) + (g1 - g1 == 0.0 ? 0.0 : 1.0);  // entries that are not finite
  }
};
struct KAdamStats {
  const double* part;
  const double* val;
  int npart;
  double* out;  // [B][3]: objective, |u_v|^2, 1 when every gradient entry is finite
  CHMC_HD void operator()(int c) const {
    double sq = 0.0, bad = 0.0;
    for (int j = 0; j < npart; ++j) sq += part[((size_t)c * npart + j) * 2], bad += part[((size_t)c * npart + j) * 2 + 1];
    out[c * 3] = val[c];
    out[c * 3 + 1] = sq;
    out[c * 3 + 2] = bad == 0.0 ? 1.0 : 0.0;
  }
};
struct KAdamUpdate {
  double* uv;
  double* m;
  double* v;
  const double* g;
  const double* coef;  // [B][2]: 1 / (1 - b2^t), lr / (1 - b1^t) (0: the chain's parameters stay)
  int n;
  double b1, b2, eps;
  CHMC_HD void operator()(int tid) const {
    const int c = tid / n;
    double gg = g[tid];
    if (!(gg - gg == 0.0)) gg = 0.0;
    const double mn = b1 * m[tid] + (1.0 - b1) * gg;
    const double vn = b2 * v[tid] + (1.0 - b2) * gg * gg;
    m[tid] = mn, v[tid] = vn;
    const double lr = coef[c * 2 + 1];
    if (lr != 0.0) uv[tid] -= lr * (mn / (sqrt(vn * coef[c * 2]) + eps));
  }
};

}  // namespace chmc
