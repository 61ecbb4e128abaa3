// HIP backend primitives used by chmc_api.inc: one non-blocking stream per process, every kernel is the
// generic 1-D launcher below instantiated with one functor of chmc_core.h.
#pragma once
#include <hip/hip_runtime.h>

// one stream per host thread that drives a context (contexts driven from different threads run concurrently)
// g_stream is the stream every primitive below enqueues on: the main stream, or -- inside a leapfrog step that is run
// as two half-batches (chmc_api.inc) -- one of the two half-batch streams selected with use_stream()
static thread_local hipStream_t g_stream = nullptr;
static thread_local hipStream_t g_streams[3] = {nullptr, nullptr, nullptr};  // main, half 0, half 1
static thread_local int g_device = -1;
static hipError_t g_first_err = hipSuccess;

static inline void note(hipError_t e) {
  if (e != hipSuccess && g_first_err == hipSuccess) g_first_err = e;
}
static int dev_set(int device) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return -1;
  }
  return 0;
}
static int dev_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_err = std::string("no HIP device available (") + hipGetErrorString(e) + ")";
    return -1;
  }
  if (device < 0 || device >= n) {
    g_err = "device ordinal out of range";
    return -1;
  }
  if (dev_set(device)) return -1;
  if (!g_streams[0] || g_device != device) {
    for (int i = 0; i < 3; ++i) {
      e = hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking);
      if (e != hipSuccess) {
        g_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        return -1;
      }
    }
    g_device = device;
  }
  g_stream = g_streams[0];
  g_first_err = hipSuccess;
  return 0;
}
static int dev_num_cus() {
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, g_device) != hipSuccess || n <= 0) n = 256;
  return n;
}
static void* dev_alloc(size_t bytes) {
  void* p = nullptr;
  note(hipMalloc(&p, bytes ? bytes : 8));
  return p;
}
static void dev_free(void* p) {
  if (p) (void)hipFree(p);
}
static void dev_zero(void* p, size_t bytes) { note(hipMemsetAsync(p, 0, bytes, g_stream)); }
static void h2d(void* d, const void* h, size_t bytes) {
  note(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_stream));
  note(hipStreamSynchronize(g_stream));
}
static void d2h(void* h, const void* d, size_t bytes) {
  note(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_stream));
  note(hipStreamSynchronize(g_stream));
}
// asynchronous read-back of one int (the Newton loop's active-chain count): the copy and an event are queued behind
// the producing kernel, the host keeps enqueueing the next iteration and only then waits for the event
static thread_local int* g_poll_host = nullptr;  // pinned, 4 slots (2 per half-batch)
static thread_local hipEvent_t g_poll_ev[4];
static void poll_begin(int slot, const int* d) {
  if (!g_poll_host) {
    note(hipHostMalloc((void**)&g_poll_host, 4 * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) note(hipEventCreateWithFlags(&g_poll_ev[i], hipEventDisableTiming));
  }
  note(hipMemcpyAsync(g_poll_host + slot, d, sizeof(int), hipMemcpyDeviceToHost, g_stream));
  note(hipEventRecord(g_poll_ev[slot], g_stream));
}
static int poll_end(int slot) {
  note(hipEventSynchronize(g_poll_ev[slot]));
  return g_poll_host[slot];
}
// A per-chain functor launch whose LAST workgroup publishes a device counter (filled by the functor's atomics) straight into
// the pinned poll slot: the count of a Newton round reaches the host without a copy packet behind the check (one launch
// gap and a 4 us blit kernel less per round).  Every thread waits for its own atomics before the workgroup barrier; the
// last workgroup (ticket) reads the finished counter with an atomic and stores it with system scope.
template <class F>
__global__ void __launch_bounds__(256) k_run_publish(F f, int n, int* counter, int* host_out, unsigned* ticket) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid < n) f(tid);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    if (t == gridDim.x - 1) {
      const int v = atomicAdd(counter, 0);
      *ticket = 0u;
      __hip_atomic_store(host_out, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static void d2d(void* dst, const void* src, size_t bytes) {
  note(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
}
// ---- two half-batches on two streams (chmc_leapfrog_step): fork / join against the main stream, and a ping-pong of
// events that keeps the halves' latency-bound forward scans out of phase (half B's scan waits for half A's, A's next
// one for B's), so that one half's scan and small kernels run under the other half's throughput-bound sweeps
static const int kStagRing = 16;
static thread_local hipEvent_t g_fork_ev = nullptr, g_join_ev[2], g_stag_ev[2][kStagRing];
static thread_local int g_stag_n[2] = {0, 0};
static thread_local int g_cur_half = -1;
static void streams_init_events() {
  if (g_fork_ev) return;
  note(hipEventCreateWithFlags(&g_fork_ev, hipEventDisableTiming));
  for (int h = 0; h < 2; ++h) {
    note(hipEventCreateWithFlags(&g_join_ev[h], hipEventDisableTiming));
    for (int i = 0; i < kStagRing; ++i) note(hipEventCreateWithFlags(&g_stag_ev[h][i], hipEventDisableTiming));
  }
}
static void use_stream(int half) {  // -1: main stream
  g_cur_half = half;
  g_stream = g_streams[half + 1];
}
static void streams_fork() {
  streams_init_events();
  note(hipEventRecord(g_fork_ev, g_streams[0]));
  for (int h = 0; h < 2; ++h) note(hipStreamWaitEvent(g_streams[h + 1], g_fork_ev, 0));
  g_stag_n[0] = g_stag_n[1] = 0;
}
static void streams_join() {
  for (int h = 0; h < 2; ++h) {
    note(hipEventRecord(g_join_ev[h], g_streams[h + 1]));
    note(hipStreamWaitEvent(g_streams[0], g_join_ev[h], 0));
  }
  use_stream(-1);
}
static const bool g_no_stagger = false;
static void stagger_wait() {  // before a forward scan of the current half: the other half's latest scan must be done
  if (g_cur_half < 0 || g_no_stagger) return;
  const int o = g_cur_half ^ 1;
  if (g_stag_n[o] > 0) note(hipStreamWaitEvent(g_stream, g_stag_ev[o][(g_stag_n[o] - 1) % kStagRing], 0));
}
static void stagger_record() {
  if (g_cur_half < 0 || g_no_stagger) return;
  const int h = g_cur_half;
  note(hipEventRecord(g_stag_ev[h][g_stag_n[h] % kStagRing], g_stream));
  g_stag_n[h]++;
}
static int dev_sync() {
  note(hipStreamSynchronize(g_stream));
  if (g_first_err != hipSuccess) {
    g_err = std::string("HIP error: ") + hipGetErrorString(g_first_err);
    g_first_err = hipSuccess;
    return -1;
  }
  return 0;
}

template <class F>
__global__ void __launch_bounds__(256) k_run(F f, int n) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < n) f(tid);
}
// optional per-launch timing with HIP events on g_stream (bench.py's roofline figure)
#include <vector>
struct ProfRec {
  hipEvent_t a, b;
  int cls;
};
static unsigned g_prof_mask = 0;  // bit per kernel class
static int g_prof_stride = 1;     // events bracket every g_prof_stride-th launch of a profiled class
static long long g_prof_seen[16];
// An event pair costs the stream ~2 x 15 us of dispatch bubbles (the command processor drains before it writes a
// time stamp), so timing every launch of a class inside a timed region slows the region itself by several per cent;
// sampling keeps the measurement live with a fraction of that cost.
static inline bool g_prof_on_for(int cls) {
  if (!((g_prof_mask >> cls) & 1u)) return false;
  return (g_prof_seen[cls]++ % g_prof_stride) == 0;
}
static std::vector<ProfRec> g_prof_pending;
static std::vector<hipEvent_t> g_prof_pool;
static double g_prof_ms[16];
static long long g_prof_n[16];
static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  note(hipEventCreate(&e));
  return e;
}
static void prof_drain() {
  if (g_prof_pending.empty()) return;
  for (int i = 0; i < 3; ++i)
    if (g_streams[i]) note(hipStreamSynchronize(g_streams[i]));
  for (auto& r : g_prof_pending) {
    float ms = 0.f;
    note(hipEventElapsedTime(&ms, r.a, r.b));
    g_prof_ms[r.cls] += ms;
    g_prof_n[r.cls] += 1;
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_pending.clear();
}
template <class F>
static void launch(F f, long n, int cls = 0) {
  if (n <= 0) return;
  const int bs = n >= 65536 ? 256 : 64;
  const unsigned grid = (unsigned)((n + bs - 1) / bs);
  ProfRec r;
  const bool prof = g_prof_on_for(cls);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = cls;
    note(hipEventRecord(r.a, g_stream));
  }
  hipLaunchKernelGGL(k_run<F>, dim3(grid), dim3(bs), 0, g_stream, f, (int)n);
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
}
// row launch for the element-wise functors: grid (column pairs / 256, B), f(c, col) handles components col, col + 1 of
// chain c.  The chain is uniform per workgroup (per-chain flags and step sizes become scalar loads) and a work item
// moves 16 bytes per operand (tools/ubench/stream.hip: 6.2 TB/s against 4.6 TB/s for one component per work item
// with the chain found by an integer division).
// Chain order of the bandwidth-bound launches.  Every vector of a step is larger than the chip's caches together, but what a
// kernel touched LAST (about the Infinity Cache's 256 MB) is still there when the next kernel starts: a launch that walks the
// chains in the opposite direction to its predecessor starts with those lines.  The grid's y index is the chain, workgroups
// are dispatched in ascending order: `rev` maps it to the chain from the other end.  The wave kernels (one wavefront per
// (chain, block) or interval) walk the work order forwards, so the launch that follows one goes backwards; consecutive row /
// column launches alternate.  Results do not depend on the order (nothing crosses chains).
// Measured at configs[1] on one box, three interleaved runs each: 48.6 k -> 49.4 k steps/s with the column launches reversed.
static thread_local int g_last_dir = 0;  // direction of the last bandwidth-bound launch: 0 forwards, 1 backwards
static inline int next_chain_direction() {
#ifdef CHMC_CHAIN_ORDER_FORWARD
  return 0;
#else
  g_last_dir ^= 1;
  return g_last_dir;
#endif
}
static inline void note_forward_launch() { g_last_dir = 0; }
template <class F>
__global__ void __launch_bounds__(256) k_rows(F f, int ncol, int rev) {
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  if (!f.active(c)) return;  // uniform per workgroup
  const int col = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (col < ncol) f(c, col);
}
template <class F>
static void launch_rows(F f, int ncol, int B, int cls = 0) {
  if (ncol <= 0 || B <= 0) return;
  ProfRec r;
  const bool prof = g_prof_on_for(cls);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = cls;
    note(hipEventRecord(r.a, g_stream));
  }
  hipLaunchKernelGGL(k_rows<F>, dim3((unsigned)(((ncol + 1) / 2 + 255) / 256), (unsigned)B), dim3(256), 0, g_stream, f, ncol,
                     next_chain_direction());
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
}
// row-sum launch: as the row launch, with `nacc` per-item accumulators that are summed over the workgroup (wave
// shuffles, then LDS) and written as one partial per (chain, workgroup): partial [B][gridDim.x][nacc].  No atomics:
// the caller adds the partials of a chain in a fixed order, so the sums are reproducible.  A work item visits
// CHMC_ROWSUM_PAIRS column pairs (stride 256 pairs: coalesced), so that the cross-lane reduction of up to 60 accumulators
// is paid once per 4 096 columns: with one pair per work item the reduction, not the memory traffic, set the time
// (KTreeLeaf at configs[1]: 1.2 ms per launch, 157 partials per chain for the single-thread second stage).
#define CHMC_ROWSUM_PAIRS 8
template <class F>
__global__ void __launch_bounds__(256) k_rowsum(F f, int ncol, int nacc, double* partial) {
  const int c = blockIdx.y;
  if (!f.active(c)) return;  // uniform per workgroup
  double acc[CHMC_ROWSUM_MAX];
#pragma unroll
  for (int a = 0; a < CHMC_ROWSUM_MAX; ++a) acc[a] = 0.0;
  for (int k = 0; k < CHMC_ROWSUM_PAIRS; ++k) {
    const int col = 2 * ((blockIdx.x * CHMC_ROWSUM_PAIRS + k) * 256 + threadIdx.x);
    if (col < ncol) f(c, col, acc);
  }
  __shared__ double sm[4][CHMC_ROWSUM_MAX];
#pragma unroll
  for (int a = 0; a < CHMC_ROWSUM_MAX; ++a) {
    if (a < nacc) {
      double v = acc[a];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][a] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nacc)
    partial[((size_t)c * gridDim.x + blockIdx.x) * nacc + threadIdx.x] =
        (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
static int rowsum_groups(int ncol) { return ((ncol + 1) / 2 + 256 * CHMC_ROWSUM_PAIRS - 1) / (256 * CHMC_ROWSUM_PAIRS); }
template <class F>
static void launch_rowsum(F f, int ncol, int B, int nacc, double* partial, int cls = 0) {
  if (ncol <= 0 || B <= 0) return;
  (void)cls;
  hipLaunchKernelGGL(k_rowsum<F>, dim3((unsigned)rowsum_groups(ncol), (unsigned)B), dim3(256), 0, g_stream, f, ncol, nacc,
                     partial);
  note(hipGetLastError());
}
// column-max launch: grid (ceil(ncol / 256), B); f(c, col) returns a bit pattern that is max-reduced per chain
template <class F>
__global__ void __launch_bounds__(256) k_colmax(F f, int ncol, int rev) {
  const int c = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  if (!f.active(c)) return;  // uniform per workgroup
  const int col = blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = 0ULL;
  if (col < ncol) v = f(c, col);
  unsigned long long* tgt = f.red(c);
  if (!tgt) return;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  __shared__ unsigned long long sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) v = sm[i] > v ? sm[i] : v;
    if constexpr (F::kFinish) {
      // Per-chain epilogue by the last workgroup of the chain (no fences: a workgroup-wide release flushes the XCD's L2,
      // measured 13x slower).  Device-scope atomics are performed at the point of coherence, and one that RETURNS a value has
      // been performed when the value arrives: the ticket is taken with an increment that depends on the value the maximum
      // returned (its top bit, always 0), so every workgroup's maximum is in place before its ticket counts.  The
      // workgroup that draws the last ticket reads the finished maximum back with an atomic and runs f.finish (the
      // chain's other data were written by earlier kernels of the stream).
      if (f.has_finish()) {
        const unsigned long long old = v ? atomicMax(tgt, v) : 0ULL;
        const unsigned t = atomicAdd(f.ticket(c), 1u + (unsigned)(old >> 63));
        if (t == gridDim.x - 1) f.finish(c, atomicMax(tgt, 0ULL));
      } else if (v) {
        atomicMax(tgt, v);
      }
    } else {
      if (v) atomicMax(tgt, v);
    }
  }
}
// One ticket word per poll slot: the check kernels of the two half-batches (CHMC_HALVES=2: slots 2 h + (round & 1)) run on
// different streams and may overlap, so they must not draw tickets from one counter (a workgroup of the wrong launch would
// then publish a partial count and the other half's count would never arrive).  Launches that share a slot are ordered by
// their stream.  The words are zeroed synchronously when they are allocated and reset by every launch's last workgroup.
static thread_local unsigned* g_publish_ticket = nullptr;
template <class F>
static void launch_publish(F f, long n, int* counter, int poll_slot) {
  if (n <= 0) return;
  if (!g_poll_host) {
    note(hipHostMalloc((void**)&g_poll_host, 4 * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) note(hipEventCreateWithFlags(&g_poll_ev[i], hipEventDisableTiming));
  }
  if (!g_publish_ticket) {
    note(hipMalloc((void**)&g_publish_ticket, 4 * sizeof(unsigned)));
    note(hipMemset(g_publish_ticket, 0, 4 * sizeof(unsigned)));
  }
  ProfRec r;
  const bool prof = g_prof_on_for(0);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = 0;
    note(hipEventRecord(r.a, g_stream));
  }
  hipLaunchKernelGGL(k_run_publish<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, f, (int)n, counter,
                     g_poll_host + poll_slot, g_publish_ticket + poll_slot);
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
  note(hipEventRecord(g_poll_ev[poll_slot], g_stream));
}
template <class F>
static void launch_colmax(F f, int ncol, int B, int cls = 0) {
  if (ncol <= 0 || B <= 0) return;
  ProfRec r;
  const bool prof = g_prof_on_for(cls);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = cls;
    note(hipEventRecord(r.a, g_stream));
  }
  hipLaunchKernelGGL(k_colmax<F>, dim3((unsigned)((ncol + 255) / 256), (unsigned)B), dim3(256), 0, g_stream, f, ncol,
                     next_chain_direction());
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
}
// wave kernels (chmc_wave.h): one (chain, block) per wavefront
template <class K, class... Args>
static void launch_wave(K kern, long nwaves, int cls, Args... args) {
  if (nwaves <= 0) return;
  ProfRec r;
  const bool prof = g_prof_on_for(cls);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = cls;
    note(hipEventRecord(r.a, g_stream));
  }
  // one wavefront per workgroup: a SIMD takes the next (chain, block) as soon as its wavefront retires instead of
  // waiting for the other three of a 256-thread workgroup (reverse sweep -3 %)
  const int wpb = 1;
  note_forward_launch();
  hipLaunchKernelGGL(kern, dim3((unsigned)((nwaves + wpb - 1) / wpb)), dim3(64 * wpb), 0, g_stream, args...);
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
}
// explicit grid / block launch (forward-scan kernel: one or two wavefronts per workgroup)
template <class K, class... Args>
static void launch_blocks(K kern, long nblocks, int bs, int cls, Args... args) {
  if (nblocks <= 0) return;
  ProfRec r;
  const bool prof = g_prof_on_for(cls);
  if (prof) {
    r.a = prof_event(), r.b = prof_event(), r.cls = cls;
    note(hipEventRecord(r.a, g_stream));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(bs), 0, g_stream, args...);
  note(hipGetLastError());
  if (prof) {
    note(hipEventRecord(r.b, g_stream));
    g_prof_pending.push_back(r);
    if (g_prof_pending.size() > 4096) prof_drain();
  }
}
extern "C" int chmc_profile_enable(int on) {
  if (g_stream) prof_drain();
  for (int i = 0; i < 16; ++i) g_prof_ms[i] = 0.0, g_prof_n[i] = 0, g_prof_seen[i] = 0;
  g_prof_mask = on == 1 ? 0xffffffffu : (unsigned)on;  // 1: every class; otherwise a bit mask of classes
  return 0;
}
extern "C" int chmc_profile_stride(int every) {
  g_prof_stride = every > 1 ? every : 1;
  return 0;
}
extern "C" int chmc_profile_get(double* ms, long long* launches) {
  if (g_stream) prof_drain();
  for (int i = 0; i < 10; ++i) {
    if (ms) ms[i] = g_prof_ms[i];
    if (launches) launches[i] = g_prof_n[i];
  }
  return 0;
}

