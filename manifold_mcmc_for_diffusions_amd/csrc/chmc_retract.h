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
//     combine    frames, Gram block, LU, core system, multipliers, mu_F       newton_comb_body<.., FACTOR> (wavefront 0)
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

__device__ __forceinline__ void wg_phase_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

template <class M, int RM, int NW>
__global__ void __launch_bounds__(64 * NW)
    k_retract_chain(Sys sy, Slots sl, Work w, int prev, int qsel, double ctol, double ptol, double dtol, int max_iters,
                    int* iters_dst) {
  constexpr int X = M::X, V = M::V;
  static_assert(RM == 16, "one 16-row block per chain");
  const int c = blockIdx.x;  // K == 1: the work order of the wave-per-block kernels is the identity
  if (c >= sy.B) return;
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
  const int s_ = sl.cur[c] ^ which;  // slot of the iterate (unless it is work.qb)
  const double* q = (qsel ? w.qb : pick(sl.q, s_)) + (size_t)c * sy.Q;
  const double* xobs = sy.xobs + (size_t)c * sy.T * X;
  const size_t toff = (size_t)c * sy.TRJ + (size_t)bd.step0 * X;
  double* traj = w.trajw + toff;
  double* out = w.cpad + (size_t)c * sy.Kmax * RM;
  const int ncol = sy.T * sy.S + sy.V0 + (sy.noisy ? sy.T : 0);
  const KUpdatePB<RM, X, V, 0, 1> upd{sy, sl, w, prev, qsel, 0, CheckArgs{}};
  for (int it = 0;; ++it) {
    // ---- constraint values and trajectory of the iterate: first iteration from the state's own trajectory, later ones
    // from the previous iterate's (this buffer).  The sweeps go on until every junction has settled; after 64 NW sweeps
    // the exact prefix has reached the end of the block whatever the guess.
    {
      const double* guess = it == 0 ? pick(sl.traj, sl.cur[c]) + toff : traj;
      double Ul[X];
      int s0;
      bool have;
      (void)fwd_par_sweeps<M, RM, NW>(sy, w, bd, q, xobs, traj, guess, out, 64 * NW + 2, it == 0 ? 2 : 1, Ul, s0, have);
      if (tid == 0)
        for (int i = bd.nrows; i < RM; ++i) out[i] = 0.0;  // padded constraint slots
    }
    wg_phase_sync();
    // ---- interval sums against the previous point's compact rows
    for (int m = wv; m < bd.nobs; m += NW) newton_ivl_body<M, false>(sy, sl, w, prev, qsel, c, 0, m, bd);
    wg_phase_sync();
    // ---- frames, Gram block, LU, core system, multipliers, mu_F; the u-part of the update, |c|_inf, |delta u|_inf
    if (wv == 0) newton_comb_body<M, RM, false, true>(sy, sl, w, prev, qsel, c, 0, bd);
    wg_phase_sync();
    // ---- q_v -= mu_F[m] . PB[s] (and the v_0 / observation-noise columns), max |delta q|
    unsigned long long r = 0ULL;
    for (int idx = tid; idx < ncol; idx += 64 * NW) {
      const unsigned long long v = upd(c, idx);
      r = v > r ? v : r;
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
      unsigned long long nb = w.ndq[c];  // (the u-part, left by the combine step)
#pragma unroll
      for (int k = 0; k < NW; ++k) nb = sMax[k] > nb ? sMax[k] : nb;
      w.ndq[c] = nb;
      const int i = it + 1;
      w.iters[c] = i;
      const double err = w.err[c], ndq = bitsd(nb);
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
    if (!sGo) break;  // (sGo is rewritten by the next iteration's check, four barriers on)
  }
}

}  // namespace chmc
