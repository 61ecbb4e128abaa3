// Whole Newton retraction of a chain in ONE kernel, one workgroup per chain: layouts with a single 16-row block per chain
// (the boarding-school SIR configuration, BASELINE configs[3]: K = 1, 14 rows, 2 800 steps).
//
// newton_projection (sde/mici_extensions.py:1065-1135) is a lax.while_loop per chain: a chain iterates exactly as long as
// IT needs, and the reference runs its chains one after the other (scripts/utils.py:351-363).  The lock-step loop of
// run_projection (chmc_api.inc) shares every launch between the chains instead, so a round costs the batch its slowest
// scan and a retraction its slowest chain's rounds: at 256 boarding-school chains 14.7 rounds of 5 launches per step for
// a mean of 8.45 iterations per chain, the time-parallel scans alone 66 % of the step at 0.02 of the HBM peak.  With one
// block per chain nothing crosses chains inside a retraction and every phase of an iteration is already shaped as
// wavefronts of one chain:
//     scan       time-parallel multiple shooting over 64 NW segments          fwd_par_sweeps<M, RM, NW>   (all wavefronts)
//     sums       one wavefront per observation interval                       newton_ivl_body             (wavefront m mod NW)
//     combine    frames, Gram block, LU, core system, multipliers, mu_F       newton_comb_wg (all threads; LU: wavefront 0)
//     update     q_v -= mu_F[m] . PB[s], max |delta q|                        KUpdatePB (functor, all threads)
//     check      the loop condition (:1119-1127), status mapping (:1462-1476) one thread
// so the workgroup walks them with workgroup barriers instead of launches, the phases hand their results on through the
// same global work arrays as the separate kernels (the chain's lines stay in this CU's L1 / the XCD's L2), and the chain
// leaves the loop when ITS condition says so.  256 chains <-> 256 CUs.  Per chain the arithmetic is that of the separate
// kernels (same device functions); the scan has 64 NW segments whatever the batch, so a chain's bits do not depend on the
// shard it runs in.
//
// Memory model: every phase ends with s_waitcnt vmcnt(0) + s_barrier (wg_phase_sync).  All wavefronts of a workgroup run on
// one CU and share its vector L1, so global data written before the barrier is visible to the workgroup after it (LLVM
// AMDGPU memory model, workgroup scope, non-tgsplit mode); hipcc only uses the (incoherent) scalar cache for memory it
// can prove unmodified by the kernel, which the stores of the other phases rule out for the work arrays.
#pragma once

namespace chmc {

#ifndef CHMC_RETRACT_WAVES
#define CHMC_RETRACT_WAVES 8  // 512 threads: two wavefronts per SIMD of the chain's CU, up to 256 registers each
#endif
// Segmentation of a chain's time-parallel scans: 64 x this many segments, on the first wavefronts of the workgroup -- and in
// the batched path of the same layouts (k_fwd_par<.., CHMC_CHAIN_SCAN_WAVES>), which must integrate the same segments to give
// the same bits.  Measured (batched path, boarding-school SIR, steps/s at 256 / 1 024 chains): 2: 65.8 k / 123.8 k, 4: 69.4 k /
// 122.3 k, 8: 55.2 k / 94.8 k -- 512 segments cost the throughput regime a quarter (junction-scan and barrier work grows with
// the wavefronts, the exact prefix of a diverging chain's scan advances one segment per sweep); per-chain kernel: see DESIGN 4.2.
#ifndef CHMC_CHAIN_SCAN_WAVES
#define CHMC_CHAIN_SCAN_WAVES 4
#endif

// -DCHMC_RETRACT_PROF: thread 0 of every workgroup adds the 100 MHz ticks of each phase to work.nfallback[48 ..] (diagnostic
// build only, tools/retract_prof.py): [48] scan [49] sums [50] combine [51] update + check [52] iterations [53] retractions
#ifdef CHMC_RETRACT_PROF
#define CHMC_RPROF(slot)                                                   \
  do {                                                                     \
    if (tid == 0) {                                                        \
      const long long t1_ = wall_clock64();                                \
      atomicAdd(w.nfallback + (slot), (int)(t1_ - t0_));                   \
      t0_ = t1_;                                                           \
    }                                                                      \
  } while (0)
#else
#define CHMC_RPROF(slot) \
  do {                   \
  } while (0)
#endif

// The chain index as a value the optimiser cannot see through: every phase of the per-chain kernels derives its addresses
// from its own opaque copy, so that loop-invariant code motion does not hoist the address arithmetic of ALL phases to the
// top of the step / iteration loop, where it would stay live through every hot loop (measured: 177 scratch instructions
// in the scan's sweep loop of the fused kernel against 19 with the copies).
__device__ __forceinline__ int opaque_u(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ void wg_phase_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// The combine step of a Newton iteration (newton_comb_body<.., FACTOR>) by the whole workgroup: the per-interval frames no
// longer walk the intervals one after the other with everything in one wavefront (14 intervals x 5 LDS round trips) --
//   (a) the interval sums and the previous point's frames go to LDS;
//   (b) one thread per row carries its adjoint row back through the interval products Pt[m] and leaves the frame LamF[m][i]
//       of every interval (the only sequential part: nobs vector-matrix products of size X);
//   (c) Ys[m][j] = Ss[m] LFprev[m][j]^T for every (m, j);  (d) the Gram block D[i][j] = sum_m LamF[m][i] . Ys[m][j] and the
//       dc/dz rows sum_m LamF[m][i] Ws[m], one thread per entry, m descending and the components ascending as in
//       newton_comb_body (same sums, same order);  (e) v_0 columns, observation-noise diagonal, identity padding, dc/du rows;
//   (f) the block's LU, the chain's core system, the multipliers and mu_F on 16 lanes of wavefront 0 (newton_factor16).
// The Cholesky factorisation of a 16-row Gram block and everything KStateFactor derives from it, on 16 lanes of one wavefront
// (lanes 0 .. 15 of the caller: row r per lane; the other lanes run along on identity rows): L (facD: lower triangle, zeros
// above, the storage chol_lower leaves), the block's log-determinant share, E = D^-1 dc/du (forward / backward substitution with
// L in LDS), C_b = (dc/du)^T E.  The factor itself is bitwise chol_lower<16> (same subtraction order); a single thread spends
// 150 us on it with the 16 x 16 matrix in scratch memory (KStateFactor: fine as one of 256 lanes of a batched launch, not
// on a chain's critical path).
template <class M, int RM>
__device__ __forceinline__ void state_factor16(const Sys& sy, const Slots& sl, const Work& w, int which, int c, bool act,
                                               const double* Dsrc, const double* Jusrc, double* Lsh /* LDS, RM x RM */) {
  static_assert(RM == 16, "rows over 16 lanes");
  constexpr int U = M::U;
  const int lane = threadIdx.x & 63, r = lane & 15;
  const int s = sl.cur[c] ^ which;
  const size_t cb = (size_t)c * sy.Kmax;
  double l[RM], e[U], ju[U];
#pragma unroll
  for (int k = 0; k < RM; ++k) l[k] = act ? Dsrc[r * RM + k] : (k == r ? 1.0 : 0.0);
#pragma unroll
  for (int d = 0; d < U; ++d) ju[d] = act ? Jusrc[r * U + d] : 0.0, e[d] = ju[d];
  double ld = 0.0;
#pragma unroll
  for (int j = 0; j < RM; ++j) {
    double t = l[j];
#pragma unroll
    for (int k = 0; k < RM; ++k)
      if (k < j) t -= l[k] * readlane_d(l[k], j);
    const double d = sqrt(readlane_d(t, j));
    ld += log(fabs(d));
    const double inv = 1.0 / d;
    l[j] = r == j ? d : (r > j ? t * inv : 0.0);
  }
  if (lane < 16) {  // (the block's own lanes; the others hold identity rows)
#pragma unroll
    for (int k = 0; k < RM; ++k) Lsh[r * RM + k] = l[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // E = L^-T L^-1 dc/du, column-oriented: row k is final, then every other row takes its multiple of it
#pragma unroll
  for (int k = 0; k < RM; ++k) {
    const double lkk = readlane_d(l[k], k);
#pragma unroll
    for (int d = 0; d < U; ++d) {
      if (r == k) e[d] = e[d] / lkk;
      const double bk = readlane_d(e[d], k);
      if (r > k) e[d] -= l[k] * bk;
    }
  }
#pragma unroll
  for (int k = RM - 1; k >= 0; --k) {
    const double lkk = readlane_d(l[k], k);
    const double lkr = Lsh[k * RM + r];  // L[k][r], r < k
#pragma unroll
    for (int d = 0; d < U; ++d) {
      if (r == k) e[d] = e[d] / lkk;
      const double bk = readlane_d(e[d], k);
      if (r < k) e[d] -= lkr * bk;
    }
  }
  double Cb[U * U];
#pragma unroll
  for (int a = 0; a < U; ++a)
#pragma unroll
    for (int d = 0; d < U; ++d) Cb[a * U + d] = row16_sum(ju[a] * e[d]);
  if (act) {
    double* fd = pick(sl.facD, s) + cb * RM * RM + r * RM;
#pragma unroll
    for (int k = 0; k < RM; ++k) fd[k] = l[k];
    double* E = pick(sl.E, s) + cb * RM * U + r * U;
#pragma unroll
    for (int d = 0; d < U; ++d) E[d] = e[d];
    if (r == 0) {
      pick(sl.ldb, s)[cb] = ld;
#pragma unroll
      for (int i = 0; i < U * U; ++i) w.Cb[cb * U * U + i] = Cb[i];
    }
  }
}

// STATE: the combine step of the state evaluation of slot `prev` (newton_comb_body<.., STATE>): the point's own frames (written to
// Slots::LF), the symmetric Gram block (also to work.Dw), the rows' v_0 columns, the slot's dc/du rows and work.zbP; then the
// block's Cholesky factor and what follows from it on 16 lanes (state_factor16) instead of the LU of a Newton iteration.
template <class M, int RM, int NW, bool STATE = false>
__device__ __forceinline__ void newton_comb_wg(const Sys& sy, const Slots& sl, const Work& w, int prev, int qsel, int c,
                                               const BlockDesc& bd) {
  constexpr int X = M::X, Z = M::Z, U = M::U, V0 = M::V0, NT = 64 * NW;
  constexpr int NI = CHMC_IVL_N(X, Z);
  constexpr int MO = RM;  // observation intervals of a block with at most RM rows
  __shared__ double Iv[MO][NI], LFp[STATE ? 1 : MO][RM * X], LamF[MO + 1][RM * X], Ys[MO][RM * X], Dl[RM * RM], zl[RM * Z],
      JuS[RM * U];
  static_assert(MO * RM * X >= RM * RM, "the Cholesky factor borrows Ys");
  const int tid = threadIdx.x;
  const int sp = sl.cur[c] ^ prev;
  const size_t cb = (size_t)c * sy.Kmax;
  const int S = sy.S, NV = sy.NV, nobs = bd.nobs;
  const double* q = (STATE ? pick(sl.q, sp) : (qsel ? w.qb : pick(sl.q, sp ^ 1))) + (size_t)c * sy.Q;
  const double* traj = (STATE ? pick(sl.traj, sp) : w.trajw) + (size_t)c * sy.TRJ + (size_t)bd.step0 * X;
  double* Jr = pick(sl.Jv, sp) + (size_t)c * RM * NV;      // (STATE: the v_0 columns are written)
  double* LFr = pick(sl.LF, sp) + cb * sy.NOBS * RM * X;   // (STATE: written)
#ifdef CHMC_COMB_PROF
  const long long tc0_ = wall_clock64();
#endif
  // (a)
  for (int e = tid; e < nobs * NI; e += NT) Iv[e / NI][e % NI] = w.ivl[cb * sy.NOBS * NI + e];
  if constexpr (!STATE)
    for (int e = tid; e < nobs * RM * X; e += NT) LFp[e / (RM * X)][e % (RM * X)] = LFr[e];
  __syncthreads();
  // (b) thread i: the adjoint row i from the end of the block back to its start; LamF[m] = the rows at the END of
  // interval m (after the rows that start there have been injected), LamF[MO] = the rows at the block's start
  if (tid < RM) {
    const int i = tid;
    double row[X], og[X];
#pragma unroll
    for (int a = 0; a < X; ++a) row[a] = 0.0, og[a] = 0.0;
    // (the observation row's gradient is fetched before the walk: inside it, the one thread's global load of interval m stalled
    // the whole wavefront once per interval -- 14 round trips in sequence, two thirds of steps (a)-(e) of a Newton iteration)
    if (i < bd.ny && i < nobs) M::obs_grad(traj + (size_t)(i + 1) * S * X, og);
    for (int m = nobs - 1; m >= 0; --m) {
      if (m < bd.ny && i == m) {
#pragma unroll
        for (int a = 0; a < X; ++a) row[a] = og[a];
      }
      if (m == nobs - 1 && !bd.last && i >= bd.ny && i < bd.ny + X) {
#pragma unroll
        for (int a = 0; a < X; ++a) row[a] = (a == i - bd.ny) ? 1.0 : row[a];
      }
#pragma unroll
      for (int a = 0; a < X; ++a) LamF[m][i * X + a] = row[a];
      double nr[X];
#pragma unroll
      for (int d = 0; d < X; ++d) {
        double tt = 0.0;
#pragma unroll
        for (int a = 0; a < X; ++a) tt += row[a] * Iv[m][X * X + X * Z + a * X + d];
        nr[d] = tt;
      }
#pragma unroll
      for (int a = 0; a < X; ++a) row[a] = nr[a];
    }
#pragma unroll
    for (int a = 0; a < X; ++a) LamF[MO][i * X + a] = row[a];
  }
  if constexpr (STATE) __syncthreads();  // (the point's own frames are the right-hand factor)
  // (c)
  for (int e = tid; e < nobs * RM * X; e += NT) {
    const int m = e / (RM * X), r = e - m * RM * X;
    const int jj = r / X, a = r - jj * X;
    double tt = 0.0;
#pragma unroll
    for (int a2 = 0; a2 < X; ++a2) tt += Iv[m][a * X + a2] * (STATE ? LamF[m][jj * X + a2] : LFp[m][jj * X + a2]);
    Ys[m][r] = tt;
    if constexpr (STATE) LFr[e] = LamF[m][r];  // the frame of interval m (Slots::LF)
  }
  __syncthreads();
  // (d) + (e)  (RM RM + RM Z work items: more than the 256 threads of the four-wavefront form)
  for (int t = tid; t < RM * RM + RM * Z; t += NT) {
  if (t < RM * RM) {
    const int i = t / RM, jj = t - i * RM;
    double tt = 0.0;
    for (int m = nobs - 1; m >= 0; --m) {
#pragma unroll
      for (int a = 0; a < X; ++a) tt += LamF[m][i * X + a] * Ys[m][jj * X + a];
    }
    if (bd.first) {  // x_0 = generate_x_0(z, v_0): the v_0 columns against the previous point's stored v_0 columns
      double dz[X * Z], dv0[X * V0];
      M::gx0_jac(dz, dv0);
      for (int d = 0; d < V0; ++d) {
        double j0 = 0.0, j1 = 0.0;
        for (int a = 0; a < X; ++a) j0 += LamF[MO][i * X + a] * dv0[a * V0 + d], j1 += LamF[MO][jj * X + a] * dv0[a * V0 + d];
        tt += j0 * (STATE ? j1 : Jr[(size_t)jj * NV + d]);
        if (STATE && jj == 0) Jr[(size_t)i * NV + d] = j0;  // the rows' v_0 columns
      }
    }
    if (i == jj) {  // noise term on the observation rows (dc_dn_l * dc_dn_r, :772-791), identity padding
      const double sg_ = sy.noisy ? sigma_at(sy, q) : 0.0;
      if (sy.noisy && i < bd.ny) tt += sg_ * sigma_at(sy, pick(sl.q, sp) + (size_t)c * sy.Q);
      if (i >= bd.nrows) tt = 1.0;
    }
    Dl[t] = tt;
    if constexpr (STATE) w.Dw[cb * RM * RM + t] = tt;
  } else {
    const int e = t - RM * RM;
    const int i = e / Z, mz = e - i * Z;
    double tt = 0.0;
    for (int m = nobs - 1; m >= 0; --m) {
#pragma unroll
      for (int a = 0; a < X; ++a) tt += LamF[m][i * X + a] * Iv[m][X * X + a * Z + mz];
    }
    if (bd.first) {
      double dz[X * Z], dv0[X * V0];
      M::gx0_jac(dz, dv0);
      for (int a = 0; a < X; ++a) {
        double dzs = 0.0;
#pragma unroll
        for (int ee = 0; ee < X * Z; ++ee) dzs = ee == a * Z + mz ? dz[ee] : dzs;
        tt += LamF[MO][i * X + a] * dzs;
      }
    }
    zl[e] = tt;
    if constexpr (STATE) w.zbP[cb * RM * Z + e] = tt;
  }
  }
  __syncthreads();
  static_assert(RM * U <= NT && NT >= 64, "one thread per dc/du entry");
  if (tid < RM * U) {  // dc/du rows of the iterate through generate_z'(u)
    double G[Z * Z];
    M::gz_jac(q, G);
    const int i = tid / U, d = tid - i * U;
    double tt = 0.0;
    if (d < Z) {
      for (int mz = 0; mz < Z; ++mz) {
        double gs = 0.0;
#pragma unroll
        for (int ee = 0; ee < Z * Z; ++ee) gs = ee == mz * Z + d ? G[ee] : gs;
        tt += zl[i * Z + mz] * gs;
      }
    } else {
      tt = i < bd.ny ? sigma_at(sy, q) * q[sy.U + sy.NV + bd.obs0 + i] : 0.0;
    }
    JuS[tid] = tt;
    if constexpr (STATE) pick(sl.JuP, sp)[cb * RM * U + tid] = tt;
  }
  __syncthreads();
#ifdef CHMC_COMB_PROF  // (diagnostic: the combine of a Newton iteration, steps (a)-(e) | (f), 10 ns ticks in work.nfallback[54 | 55])
  long long tc1_ = 0;
  if (!STATE && tid == 0) tc1_ = wall_clock64(), atomicAdd(w.nfallback + 54, (int)(tc1_ - tc0_));
#endif
  // (f)
  if (tid < 64) {
    if constexpr (STATE) state_factor16<M, RM>(sy, sl, w, prev, c, tid < 16, Dl, JuS, &Ys[0][0]);  // (Ys is free: L goes there)
    else newton_factor16<M, RM, true, true>(sy, sl, w, prev, qsel, c, 0, tid < 16, Dl, JuS, &LFp[0][0]);  // (mu_F from the LDS copy of the frames)
  }
#ifdef CHMC_COMB_PROF
  if (!STATE && tid == 0) atomicAdd(w.nfallback + 55, (int)(wall_clock64() - tc1_));
#endif
}

// J p and J pg (k_jw_pb<.., TWO>) of the chain's block by the whole workgroup: the per-interval sums y_m = sum_s PB[s] w_s go to
// the wavefronts (interval m to wavefront m mod NW), the frames are applied by 16 threads in jw_pb_body's order (m ascending).
template <int RM, int X, int V, int NW>
__device__ __forceinline__ void jw_pb_wg(const Sys& sy, const Slots& sl, const Work& w, int which, bool minv, int c,
                                         const BlockDesc& bd) {
  __shared__ double sY[RM][2][X];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (!w.ok[c]) return;  // (uniform)
  const int s = sl.cur[c] ^ which;
  const double* vct = pick(sl.p, s) + (size_t)c * sy.Q;
  const double* vct2 = pick(sl.pg, s) + (size_t)c * sy.Q;
  const size_t cb = (size_t)c * sy.Kmax;
  const double* PB = pick(sl.PB, s) + ((size_t)c * sy.T * sy.S + bd.step0) * (X * V);
  const double* LF = pick(sl.LF, s) + cb * sy.NOBS * RM * X;
  const double* wv1 = vct + sy.U + sy.V0 + (size_t)bd.step0 * V;
  const double* wv2 = vct2 + sy.U + sy.V0 + (size_t)bd.step0 * V;
  for (int m = wv; m < bd.nobs; m += NW) {
    double y[X], y2[X];
#pragma unroll
    for (int a = 0; a < X; ++a) y[a] = 0.0, y2[a] = 0.0;
    for (int k = m * sy.S + lane; k < (m + 1) * sy.S; k += 64) {
      double pb[X * V], x[V], x2[V];
      const double* src = PB + (size_t)k * (X * V);
#pragma unroll
      for (int e = 0; e < X * V; ++e) pb[e] = ld_stream(src + e);
#pragma unroll
      for (int d = 0; d < V; ++d) x[d] = wv1[(size_t)k * V + d], x2[d] = wv2[(size_t)k * V + d];
#pragma unroll
      for (int a = 0; a < X; ++a)
#pragma unroll
        for (int d = 0; d < V; ++d) {
          y[a] += pb[a * V + d] * x[d];
          y2[a] += pb[a * V + d] * x2[d];
        }
    }
#pragma unroll
    for (int a = 0; a < X; ++a) {
      double v = y[a], v2 = y2[a];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64), v2 += __shfl_xor(v2, o, 64);
      if (lane == 0) sY[m][0][a] = v, sY[m][1][a] = v2;
    }
  }
  __syncthreads();
  if (tid < RM) {
    const int i = tid;
    double a = 0.0, a2 = 0.0;
    if (i < bd.nrows) {
      for (int m = 0; m < bd.nobs; ++m) {
        const double* lf = LF + ((size_t)m * RM + i) * X;
#pragma unroll
        for (int e = 0; e < X; ++e) {
          const double f = lf[e];
          a += f * sY[m][0][e];
          a2 += f * sY[m][1][e];
        }
      }
      if (bd.first) {  // v_0 columns: from the stored rows
        const double* Jv = pick(sl.Jv, s) + (size_t)c * RM * sy.NV + (size_t)i * sy.NV;
        for (int d = 0; d < sy.V0; ++d) a += Jv[d] * vct[sy.U + d], a2 += Jv[d] * vct2[sy.U + d];
      }
      const double* ju = pick(sl.JuP, s) + (cb * RM + i) * sy.U;
      for (int d = 0; d < sy.U; ++d) a += ju[d] * (minv ? metric_inv_u(sy, vct, d) : vct[d]);
      const double sg = sy.noisy ? sigma_at(sy, pick(sl.q, s) + (size_t)c * sy.Q) : 0.0;
      if (sy.noisy && i < bd.ny) a += sg * vct[sy.U + sy.NV + bd.obs0 + i];
      for (int d = 0; d < sy.U; ++d) a2 += ju[d] * (minv ? metric_inv_u(sy, vct2, d) : vct2[d]);
      if (sy.noisy && i < bd.ny) a2 += sg * vct2[sy.U + sy.NV + bd.obs0 + i];
    }
    w.cpad[cb * RM + i] = a;
    w.cpad2[cb * RM + i] = a2;
  }
}

// gld_ivl_prologue_body (chmc_wave.h) for the chain's one block by the whole workgroup.  The wavefront form walks the
// observation intervals one after the other with global loads and three wavefront barriers per interval (83 us per step of
// a boarding-school chain, measured) although only the row tangents xd(t_m) form a recursion: here M LF[m] and the interval
// matrices of EVERY interval are staged in LDS by all threads, 48 lanes run the recursion from LDS, and C1 / C2 / Q0 / the
// terminal tangents of every interval follow in parallel.  Every entry is the same sum in the same order as in the wavefront
// form (the batched path keeps k_gld_ivl_prologue: same bits).
template <class M, int RM, int NW>
__device__ __forceinline__ void gld_ivl_prologue_wg(const Sys& sy, const Slots& sl, const Work& w, int which, int c,
                                                    const BlockDesc& bd, double* Mb /* LDS, RM x RM */) {
  constexpr int X = M::X, Z = M::Z, V0 = M::V0, NT = 64 * NW;
  constexpr int NI = CHMC_IVL_N(X, Z), NC = CHMC_GCQ_N(X, Z), MO = RM;
  static_assert(RM * X <= 64, "the row tangents live in one wavefront");
  __shared__ double zd[RM * Z], MLFa[MO][RM * X], Iva[MO][NI], xda[MO + 1][RM * X];
  const int tid = threadIdx.x;
  if (!w.ok[c]) return;  // (uniform)
  const int s_ = sl.cur[c] ^ which;
  const size_t cb = (size_t)c * sy.Kmax;
  const int NV = sy.NV, nobs = bd.nobs;
  const double* Jv = pick(sl.Jv, s_) + (size_t)c * RM * NV;
  const double* LFr = pick(sl.LF, s_) + cb * sy.NOBS * RM * X;
  for (int i = tid; i < RM * RM; i += NT) Mb[i] = w.gMb[cb * RM * RM + i];
  for (int i = tid; i < RM * Z; i += NT) zd[i] = w.gzd[cb * RM * Z + i];
  for (int e = tid; e < nobs * NI; e += NT) Iva[e / NI][e % NI] = w.ivl[cb * sy.NOBS * NI + e];
  __syncthreads();
  for (int e = tid; e < nobs * RM * X; e += NT) {  // M LF[j] of every interval
    const int j = e / (RM * X), l = e - j * RM * X;
    const int i = l / X, a = l - i * X;
    double t = 0.0;
    for (int jj = 0; jj < RM; ++jj) t += Mb[i * RM + jj] * LFr[((size_t)j * RM + jj) * X + a];
    MLFa[j][l] = t;
  }
  if (tid < RM * X) {  // the row tangents at the block's start
    const int lane = tid;
    const int i = lane / X, a = lane - i * X;
    double t = 0.0;
    if (bd.first) {
      double dz[X * Z], dv0[X * V0];
      M::gx0_jac(dz, dv0);
      for (int mz = 0; mz < Z; ++mz) {
        double dzs = 0.0;
#pragma unroll
        for (int e2 = 0; e2 < X * Z; ++e2) dzs = e2 == a * Z + mz ? dz[e2] : dzs;
        t += dzs * zd[i * Z + mz];
      }
      for (int d = 0; d < V0; ++d) {
        double wv = 0.0, dvs = 0.0;
        for (int jj = 0; jj < RM; ++jj) wv += Mb[i * RM + jj] * Jv[(size_t)jj * NV + d];
#pragma unroll
        for (int e2 = 0; e2 < X * V0; ++e2) dvs = e2 == a * V0 + d ? dv0[e2] : dvs;
        t += dvs * wv;
      }
    }
    xda[0][lane] = t;
  }
  __syncthreads();
  if (tid < 64) {  // the recursion over the intervals: xd(t_{j+1}) from xd(t_j), one wavefront, everything in LDS
    const int lane = tid;
    for (int j = 0; j < nobs; ++j) {
      const double* xds = xda[j];
      const double* MLFs = MLFa[j];
      const double* Iv = Iva[j];
      double nx = 0.0;
      if (lane < RM * X) {
        const int i = lane / X, a = lane - i * X;
#pragma unroll
        for (int d = 0; d < X; ++d) nx += Iv[X * X + X * Z + a * X + d] * xds[i * X + d] + Iv[a * X + d] * MLFs[i * X + d];
        for (int mz = 0; mz < Z; ++mz) nx += Iv[X * X + a * Z + mz] * zd[i * Z + mz];
        xda[j + 1][lane] = nx;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int e = tid; e < nobs * X * X; e += NT) {  // C1 = LF^T M LF, Q0 = LF^T xd(t_j)
    const int j = e / (X * X), lane = e - j * X * X;
    const int a1 = lane / X, a2 = lane - a1 * X;
    double c1 = 0.0, q0 = 0.0;
    for (int i = 0; i < RM; ++i) {
      const double lf = LFr[((size_t)j * RM + i) * X + a1];
      c1 += lf * MLFa[j][i * X + a2];
      q0 += lf * xda[j][i * X + a2];
    }
    double* Cj = w.gcq + (cb * sy.NOBS + j) * NC;
    Cj[lane] = c1;
    Cj[X * X + X * Z + lane] = q0;
  }
  for (int e = tid; e < nobs * X * Z; e += NT) {  // C2 = LF^T zd
    const int j = e / (X * Z), lane = e - j * X * Z;
    const int a1 = lane / Z, mz = lane - a1 * Z;
    double c2 = 0.0;
    for (int i = 0; i < RM; ++i) c2 += LFr[((size_t)j * RM + i) * X + a1] * zd[i * Z + mz];
    w.gcq[(cb * sy.NOBS + j) * NC + X * X + lane] = c2;
  }
  for (int e = tid; e < nobs * X; e += NT) {  // terminal tangents of the observation rows: row j at the end of interval j
    const int j = e / X, a = e - j * X;
    if (j < bd.ny) w.gxdt[(cb * RM + j) * X + a] = xda[j + 1][j * X + a];
  }
}

// KSymBlk (t_b = D_b^-1 v_b against the block's Cholesky factor, s_b = Ju_b^T t_b) for the chain's 16-row block by ONE
// wavefront instead of one thread (whose 16 x 16 factor lived in scratch memory: two of them were most of the 93 us of a
// step's "J p + core solves").  Bitwise cho_solve<16, 1>: the forward substitution runs column by column over the lanes (lane i
// subtracts L_ik x_k for ascending k, the functor's order), the backward substitution -- whose sums over k > i ascend towards
// values found last -- runs as the functor's own unrolled loop, the factor read from LDS; every lane holds the result.
template <class M, int RM>
__device__ __forceinline__ void sym_blk16_wave(const Sys& sy, const Slots& sl, const Work& w, int which, int c,
                                               double* Ls /* LDS, RM x RM */) {
  static_assert(RM == 16, "rows over 16 lanes");
  constexpr int U = M::U;
  const int lane = threadIdx.x & 63;
  if (!w.ok[c]) return;  // (uniform)
  const int s = sl.cur[c] ^ which;
  const size_t cb = (size_t)c * sy.Kmax;
  const double* fd = pick(sl.facD, s) + cb * RM * RM;
  for (int i = lane; i < RM * RM; i += 64) Ls[i] = fd[i];
  double t = lane < RM ? w.cpad[cb * RM + lane] : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < RM; ++k) {
    const double xk = readlane_d(t, k) / Ls[k * RM + k];
    if (lane == k) t = xk;
    if (lane > k && lane < RM) t -= Ls[lane * RM + k] * xk;
  }
  double x[RM];
#pragma unroll
  for (int k = 0; k < RM; ++k) x[k] = readlane_d(t, k);
#pragma unroll
  for (int i = RM - 1; i >= 0; --i) {
    double tt = x[i];
#pragma unroll
    for (int k = 0; k < RM; ++k)
      if (k > i) tt -= Ls[k * RM + i] * x[k];
    x[i] = tt / Ls[i * RM + i];
  }
  if (lane < RM) {
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < RM; ++k) mine = lane == k ? x[k] : mine;
    w.tpad[cb * RM + lane] = mine;
  }
  if (lane < U) {
    const double* ju = pick(sl.JuP, s) + cb * RM * U;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < RM; ++i) acc += ju[i * U + lane] * x[i];
    w.sb[cb * U + lane] = acc;
  }
}

// The per-chain phases as kernels of their own (one workgroup per chain), for the lock-step path of the same layouts: with them
// a batched launch does the arithmetic of the per-chain kernels bit for bit, so the choice between the two execution models
// (per-chain workgroups for up to a few chains per CU, batched launches for throughput beyond that) never changes a result.
template <class M, int RM, int NW, bool STATE>
__global__ void __launch_bounds__(64 * NW) k_newton_comb_wg(Sys sy, Slots sl, Work w, int which, int qsel) {
  const int c = blockIdx.x;
  if (c >= sy.B) return;
  if (STATE ? !w.ok[c] : !newton_select(w, c, which, qsel)) return;  // (uniform over the workgroup)
  newton_comb_wg<M, RM, NW, STATE>(sy, sl, w, which, qsel, c, sy.blk[0]);
}
template <int RM, int X, int V, int NW>
__global__ void __launch_bounds__(64 * NW) k_jw_pb_wg(Sys sy, Slots sl, Work w, int which) {
  const int c = blockIdx.x;
  if (c >= sy.B) return;
  jw_pb_wg<RM, X, V, NW>(sy, sl, w, which, true, c, sy.blk[0]);
}

// The retraction of chain c by the calling workgroup (every thread calls it; returns with the chain's loop finished: status
// in work.nstat / ok / status, counts in work.iters and *iters_dst, the last iterate's trajectory in work.trajw).
template <class M, int RM, int NW>
__device__ __forceinline__ void retract_chain_body(const Sys& sy, const Slots& sl, const Work& w, int c, int prev, int qsel,
                                                   double ctol, double ptol, double dtol, int max_iters, int* iters_dst) {
  constexpr int X = M::X, V = M::V;
  static_assert(RM == 16, "one 16-row block per chain");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  __shared__ unsigned long long sMax[NW];
  __shared__ int sGo;
  // KNewtonBegin
  const bool active = w.ok[c] != 0;
  if (tid == 0) {
    w.iters[c] = 0;
    w.err[c] = -1.0;
    w.ndq[c] = 0x7ff0000000000000ULL;  // +inf
    w.nstat[c] = 0;
    w.nw[c] = 0;
    if (active && max_iters <= 0) w.nstat[c] = 1, w.ok[c] = 0, w.status[c] = 1;  // no iteration allowed: not converged
  }
  if (!active || max_iters <= 0) return;  // (uniform over the workgroup)
  const BlockDesc bd = sy.blk[0];
  const int which = prev ^ 1;
  const int ncol = sy.T * sy.S + sy.V0 + (sy.noisy ? sy.T : 0);
#ifdef CHMC_RETRACT_PROF
  long long t0_ = wall_clock64();
  if (tid == 0) atomicAdd(w.nfallback + 53, 1);
#endif
  for (int it = 0;; ++it) {
    // ---- constraint values and trajectory of the iterate: first iteration from the state's own trajectory, later ones
    // from the previous iterate's (this buffer).  The sweeps go on until every junction has settled; after 64 NW sweeps
    // the exact prefix has reached the end of the block whatever the guess.
    {
      const int ca = opaque_u(c);  // (every phase from its own copy of the chain index: see opaque_u)
      const int s_ = sl.cur[ca] ^ which;  // slot of the iterate (unless it is work.qb)
      const double* q = (qsel ? w.qb : pick(sl.q, s_)) + (size_t)ca * sy.Q;
      const size_t toff = (size_t)ca * sy.TRJ + (size_t)bd.step0 * X;
      double* traj = w.trajw + toff;
      double* out = w.cpad + (size_t)ca * sy.Kmax * RM;
      const double* guess = it == 0 ? pick(sl.traj, sl.cur[ca]) + toff : traj;
      double Ul[X];
      int s0;
      bool have;
      (void)fwd_par_sweeps<M, RM, CHMC_CHAIN_SCAN_WAVES, 0, NW>(sy, w, bd, q, sy.xobs + (size_t)ca * sy.T * X, traj, guess, out,
                                                                64 * CHMC_CHAIN_SCAN_WAVES + 2,
                                         it == 0 ? 2 : 1, Ul, s0, have);
      if (tid == 0)
        for (int i = bd.nrows; i < RM; ++i) out[i] = 0.0;  // padded constraint slots
    }
    wg_phase_sync();
    CHMC_RPROF(48);
    // ---- interval sums against the previous point's compact rows
    {
      const int cb_ = opaque_u(c);
      for (int m = wv; m < bd.nobs; m += NW) newton_ivl_body<M, false>(sy, sl, w, prev, qsel, cb_, 0, m, bd);
    }
    wg_phase_sync();
    CHMC_RPROF(49);
    // ---- frames, Gram block, LU, core system, multipliers, mu_F; the u-part of the update, |c|_inf, |delta u|_inf
    newton_comb_wg<M, RM, NW>(sy, sl, w, prev, qsel, opaque_u(c), bd);
    wg_phase_sync();
    CHMC_RPROF(50);
    // ---- q_v -= mu_F[m] . PB[s] (and the v_0 / observation-noise columns), max |delta q|
    // (Measured and rejected: two consecutive steps per work item (KUpdatePB<.., 2>): 19.6 -> 30.6 us per iteration; the loads of
    // three steps issued before the first store: 42.5 us -- every variant that holds more than one step's 15 doubles per thread
    // spills in this kernel, whose 256 registers are set by the scan and the factorisation.)
    double err0 = 0.0;
    unsigned long long nb0 = 0ULL;
    const int cu = opaque_u(c);
    if (tid == 0) err0 = w.err[cu], nb0 = w.ndq[cu];  // (left by the combine step; in flight under the update pass)
    unsigned long long r = 0ULL;
    {
      const KUpdatePB<RM, X, V, 0, 1> upd{sy, sl, w, prev, qsel, 0, CheckArgs{}};
      for (int idx = tid; idx < ncol; idx += 64 * NW) {
        const unsigned long long v = upd(cu, idx);
        r = v > r ? v : r;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(r, off, 64);
      r = o > r ? o : r;
    }
    if (lane == 0) sMax[wv] = r;
    wg_phase_sync();
    // ---- the loop condition (:1119-1127) after this iteration, status mapping of the host wrapper (:1462-1476)
    if (tid == 0) {
      unsigned long long nb = nb0;  // (the u-part, left by the combine step)
#pragma unroll
      for (int k = 0; k < NW; ++k) nb = sMax[k] > nb ? sMax[k] : nb;
      w.ndq[c] = nb;
      const int i = it + 1;
      w.iters[c] = i;
      const double err = err0, ndq = bitsd(nb);
      const bool diverged = (err > dtol) || (err != err);
      const bool converged = (err < ctol) && (ndq < ptol);
      const bool stop = i >= max_iters || diverged || converged;
      if (stop) {
        const int st = converged ? 0 : (diverged ? 2 : 1);
        w.nstat[c] = st;
        if (st) w.ok[c] = 0, w.status[c] = st;
        if (iters_dst) iters_dst[c] += i;
      }
      sGo = stop ? 0 : 1;
    }
    wg_phase_sync();
    CHMC_RPROF(51);
#ifdef CHMC_RETRACT_PROF
    if (tid == 0) atomicAdd(w.nfallback + 52, 1);
#endif
    if (!sGo) break;  // (sGo is rewritten by the next iteration's check, four barriers on)
  }
}
template <class M, int RM, int NW>
__global__ void __launch_bounds__(64 * NW, 2)  // (two wavefronts per SIMD: 256 registers, two 4-wavefront workgroups per CU)
    k_retract_chain(Sys sy, Slots sl, Work w, int prev, int qsel, double ctol, double ptol, double dtol, int max_iters,
                    int* iters_dst) {
  const int c = blockIdx.x;  // K == 1: the work order of the wave-per-block kernels is the identity
  if (c >= sy.B) return;
  retract_chain_body<M, RM, NW>(sy, sl, w, c, prev, qsel, ctol, ptol, dtol, max_iters, iters_dst);
}


// ---------------------------------------------------------------------------------------------------------------
// Whole constrained leapfrog steps of a chain -- a whole trajectory -- in one kernel, one workgroup per chain.
//
// With the retraction per chain (above) a step of the batch still waited twice for its slowest chain: a launch of
// k_retract_chain lasts as long as the chain that needs the most Newton iterations (a diverging retraction runs 10 - 30
// iterations before |c| passes the divergence tolerance; at 1 % failing chain-steps nearly every launch of 256 chains has
// one: mean launch 1.9 ms for a mean chain time of 0.9 ms), and the 21 kernels of the state evaluation and momentum
// projection between the two retractions are launch-latency work for 256 chains.  Nothing in a step crosses chains
// (scripts/utils.py:351-363 runs the chains one after the other), so the workgroup that owns the chain walks the whole
// step -- A(dt/2) + h2 flow | forward retraction | state evaluation (scan, interval sums, combine, Cholesky, chain core,
// grad log det sweeps) | momentum correction + projection | reverse flow | reverse retraction | reversibility check |
// A(dt/2) | commit (mici ConstrainedLeapfrogIntegrator.step) -- with workgroup barriers between the phases,
// and goes on to the chain's next step: n_steps[c] steps per call (the loop of an integration transition around
// integrator.step, scripts/utils.py:284-301), ended by the chain's first failed step.  A slow chain costs only itself.
// Every phase is the device function of the corresponding kernel of the lock-step path (same arithmetic per chain);
// phases hand over through the same global work arrays.
template <class F>
__device__ __forceinline__ void wg_rows(const F& f, int c, int ncol, int nt) {  // k_rows for one chain
  if (!f.active(c)) return;
  for (int col = 2 * (int)threadIdx.x; col < ncol; col += 2 * nt) f(c, col);
}
__device__ __forceinline__ void wave_sync() {  // the calling wavefront's own global / LDS accesses have completed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <class M, int RM, int NW>
__global__ void __launch_bounds__(64 * NW, 2)
    k_traj_chain(Sys sy, Slots sl, Work w, const int* n_steps, int n_steps_all, double ctol, double ptol, double dtol,
                 int max_iters, double rev_tol, int* itf, int* itb, int* n_done) {
  constexpr int X = M::X, V = M::V, NT = 64 * NW;
  const int c = blockIdx.x;
  if (c >= sy.B) return;
  const int tid = threadIdx.x, wv = tid >> 6;
  __shared__ int sOk;
  __shared__ unsigned long long sRev[NW];
  __shared__ double wgL[RM * RM];  // a 16 x 16 matrix for whichever phase needs one (interval prologue, block solves)
  const BlockDesc bd = sy.blk[0];
  const int nst = n_steps ? n_steps[c] : n_steps_all;
  const int ncol = sy.T * sy.S + sy.V0 + (sy.noisy ? sy.T : 0);
  const double kick = 0.5;
  unsigned long long rev_last = 0ULL;
  int done = 0;
  auto chain_ok = [&]() {  // work.ok[c] as seen by every thread (written by thread 0 / lane 0 of the phases before)
    wg_phase_sync();
    if (tid == 0) sOk = w.ok[c];
    wg_phase_sync();
    return sOk != 0;
  };
  if (!chain_ok()) return;  // inactive chain (status -1 from KBegin)
#ifdef CHMC_RETRACT_PROF
  long long tq_ = wall_clock64();
#define CHMC_TPROF(slot)                                      \
  do {                                                        \
    if (tid == 0) {                                           \
      const long long t1_ = wall_clock64();                   \
      atomicAdd(w.nfallback + 64 + (slot), (int)((t1_ - tq_) >> 4)); /* 160 ns units */ \
      tq_ = t1_;                                              \
    }                                                         \
  } while (0)
#ifdef CHMC_COMB_PROF
#define CHMC_TPROF2_ON 0
#else
#define CHMC_TPROF2_ON 1
#endif
// (wavefront 0's own timeline inside the state-evaluation phase: work.nfallback[54 | 55 | 63] = combine + Cholesky | chain core |
// grad-log-det preparation, 10 ns ticks since the phase's last CHMC_TPROF; the phase's remainder is the interval prologue)
#define CHMC_TPROF2(slot)                                                         \
  do {                                                                            \
    if (CHMC_TPROF2_ON && tid == 0) {                                                               \
      const long long t1_ = wall_clock64();                                       \
      atomicAdd(w.nfallback + (slot), (int)(t1_ - tq2_));                         \
      tq2_ = t1_;                                                                 \
    }                                                                             \
  } while (0)
#else
#define CHMC_TPROF(slot) \
  do {                   \
  } while (0)
#define CHMC_TPROF2(slot) \
  do {                    \
  } while (0)
#endif
  for (int step = 0; step < nst; ++step) {
    if (tid == 0) w.rev[c] = 0ULL;
    // ---- A(dt/2) for a tangent momentum (p - dt/2 pg) and the h2 flow into the proposal slot
    wg_rows(KKickFlowPg{sy, sl, w, kick}, c, sy.Q, NT);
    wg_phase_sync();
    const int c1 = opaque_u(c);
    CHMC_TPROF(0);
    // ---- retraction onto the manifold
    retract_chain_body<M, RM, NW>(sy, sl, w, c1, 0, 0, ctol, ptol, dtol, max_iters, itf);
    if (!chain_ok()) break;
    const int c2 = opaque_u(c);
    CHMC_TPROF(1);
    // ---- J, Gram factors, log det, grad log det at the new point (state_eval_core(1, true, 3), 16-row interval-parallel form)
    {
      const int s1 = sl.cur[c2] ^ 1;
      const size_t toff = (size_t)c2 * sy.TRJ + (size_t)bd.step0 * X;
      const double* q1 = pick(sl.q, s1) + (size_t)c2 * sy.Q;
      double* traj1 = pick(sl.traj, s1) + toff;
      double* out = w.cpad + (size_t)c2 * sy.Kmax * RM;
      double Ul[X];
      int s0;
      bool have;
      (void)fwd_par_sweeps<M, RM, CHMC_CHAIN_SCAN_WAVES, 0, NW>(sy, w, bd, q1, sy.xobs + (size_t)c2 * sy.T * X, traj1, w.trajw + toff, out,
                                         64 * CHMC_CHAIN_SCAN_WAVES + 2, 3, Ul, s0, have);
      if (tid == 0)
        for (int i = bd.nrows; i < RM; ++i) out[i] = 0.0;
    }
    wg_phase_sync();
    const int c3 = opaque_u(c);
    CHMC_TPROF(2);
    for (int m = wv; m < bd.nobs; m += NW) newton_ivl_body<M, true>(sy, sl, w, 1, 0, c3, 0, m, bd);
    wg_phase_sync();
    const int c4 = opaque_u(c);
    CHMC_TPROF(3);
#ifdef CHMC_RETRACT_PROF
    long long tq2_ = wall_clock64();
#endif
    newton_comb_wg<M, RM, NW, true>(sy, sl, w, 1, 0, c4, bd);  // ... and the block's Cholesky factor, E, C_b (16 lanes)
    if (wv == 0) {
      wave_sync();
      CHMC_TPROF2(54);
      if (tid == 0) KStateChain<M>{sy, sl, w, 1}(c4);
      wave_sync();
      CHMC_TPROF2(55);
      gld_prep_body<M, RM>(sy, sl, w, 1, c4, 1);
      wave_sync();
      CHMC_TPROF2(63);
    }
    wg_phase_sync();
    gld_ivl_prologue_wg<M, RM, NW>(sy, sl, w, 1, opaque_u(c), bd, wgL);
    wg_phase_sync();
    const int c5 = opaque_u(c);
    CHMC_TPROF(4);
    for (int m = wv; m < sy.NOBS; m += NW) gld_fwd_ivl_body<M, RM, false>(sy, sl, w, 1, c5 * sy.NOBS + m);
    wg_phase_sync();
    const int c6 = opaque_u(c);
    CHMC_TPROF(5);
    for (int m = wv; m < sy.NOBS; m += NW) gld_bwd_ivl_body<M, RM, 0>(sy, sl, w, 1, c6 * sy.NOBS + m);
    wg_phase_sync();
    const int c7 = opaque_u(c);
    for (int m = wv; m < sy.NOBS; m += NW) gld_bwd_ivl_body<M, RM, 1>(sy, sl, w, 1, c7 * sy.NOBS + m);
    wg_phase_sync();
    const int c8 = opaque_u(c);
    CHMC_TPROF(6);
    if (tid == 0) {
      gld_ivl_finish_body<M, RM>(sy, sl, w, 1, c8);
      KGldChain<M>{sy, sl, w, 1}(c8);
    }
    wg_phase_sync();
    const int c9 = opaque_u(c);
    CHMC_TPROF(7);
    // ---- momentum correction p -= dh2_flow_mom_dmom @ (mu / dt), pg <- dh1_dpos; then P p and pg = P dh1_dpos in one pass
    wg_rows(KMomFixInitPg{sy, sl, w, 1}, c9, sy.Q, NT);
    wg_phase_sync();
    const int c10 = opaque_u(c);
    CHMC_TPROF(8);
    jw_pb_wg<RM, X, V, NW>(sy, sl, w, 1, true, c10, bd);
    if (wv == 0) {
      wave_sync();
      Work w1 = w, w2 = w;
      w1.lampad = w.lampad2;
      w2.cpad = w.cpad2;
      sym_blk16_wave<M, RM>(sy, sl, w1, 1, c10, wgL);
      wave_sync();
      solve_chain_body<M, RM, 1, 1>(sy, sl, w1, 1, 0, 0, c10);
      wave_sync();
      sym_blk16_wave<M, RM>(sy, sl, w2, 1, c10, wgL);
      wave_sync();
      solve_chain_body<M, RM, 1, 1>(sy, sl, w2, 1, 0, 3, c10);
    }
    wg_phase_sync();
    const int c11 = opaque_u(c);
    CHMC_TPROF(9);
    {
      const KMuF<RM, X, 3> muf{sy, sl, w, 1};
      for (int e = tid; e < sy.NOBS * X; e += NT) muf(c11 * sy.NOBS * X + e);
    }
    wg_phase_sync();
    const int c12 = opaque_u(c);
    {
      const KUpdatePB<RM, X, V, 3, 1> up3{sy, sl, w, 1, 0, 0, CheckArgs{}};
      for (int idx = tid; idx < ncol; idx += NT) (void)up3(c12, idx);
    }
    wg_phase_sync();
    const int c13 = opaque_u(c);
    // ---- reverse flow from the new point into work.qb, reverse retraction along J(new point), reversibility check
    wg_rows(KFlow{sy, sl, w, 1, 1, 0, -1.0}, c13, sy.Q, NT);
    wg_phase_sync();
    const int c14 = opaque_u(c);
    CHMC_TPROF(10);
    retract_chain_body<M, RM, NW>(sy, sl, w, c14, 1, 1, ctol, ptol, dtol, max_iters, itb);
    if (!chain_ok()) break;
    const int c15 = opaque_u(c);
    CHMC_TPROF(1);
    {
      const KRevDiff rd{sy, sl, w};
      unsigned long long r = 0ULL;
      for (int idx = tid; idx < (sy.Q + 1) / 2; idx += NT) {
        const unsigned long long v = rd(c15, idx);
        r = v > r ? v : r;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(r, off, 64);
        r = o > r ? o : r;
      }
      if ((tid & 63) == 0) sRev[wv] = r;
      wg_phase_sync();
      if (tid == 0) {
#pragma unroll
        for (int k = 1; k < NW; ++k) r = sRev[k] > r ? sRev[k] : r;
        w.rev[c15] = r;
        rev_last = r;
        KRevCheck{w, rev_tol}(c15);
      }
    }
    if (!chain_ok()) break;
    const int c16 = opaque_u(c);
    // ---- A(dt/2) at the new point (momentum projected there: p - dt/2 pg), accept
    wg_rows(KKickPg{sy, sl, w, 1, 0, kick}, c16, sy.Q, NT);
    wg_phase_sync();
    const int c17 = opaque_u(c);
    if (tid == 0) KCommit{sl, w}(c17);
    ++done;
    wg_phase_sync();
    CHMC_TPROF(11);
  }
  if (tid == 0) {
    if (n_done) n_done[c] = done;
    w.rev[c] = rev_last;  // (the reverse-check distance of the last step that reached the check)
  }
}

}  // namespace chmc
