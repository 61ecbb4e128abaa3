// Micro-benchmark behind the element-wise / update kernel notes in DESIGN.md: which launch shape streams [B][Q] fp64
// vectors (B = 256 chains, Q = 80 106) at HBM rate on gfx950?  Every variant computes c = a - h[chain] * b on buffers
// far larger than the Infinity Cache, rotating over NSET buffer sets so that no launch finds its operands cached.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream.bin stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct d2 {
  double x, y;
};
// (1) the shape of the element-wise functors: one component per work item, chain from an integer division,
//     per-chain flags through dependent loads
__global__ void __launch_bounds__(256) k_flat1(const double* a, const double* b, double* c, const int* ok, const int* cur,
                                               const double* h, int Q, long n) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= n) return;
  const int ch = (int)(tid / Q);
  if (!ok[ch]) return;
  const int s = cur[ch];
  const double* aa = s ? b : a;
  const double* bb = s ? a : b;
  c[tid] = aa[tid] - h[ch] * bb[tid];
}
// (2) chain = blockIdx.y (uniform: flags become scalar loads), VEC pairs of components per work item
template <int NP, bool NT>
__global__ void __launch_bounds__(256) k_rows(const double* a, const double* b, double* c, const int* ok, const int* cur,
                                              const double* h, int Q) {
  const int ch = blockIdx.y;
  if (!ok[ch]) return;
  const int s = cur[ch];
  const double hh = h[ch];
  const double* aa = (s ? b : a) + (size_t)ch * Q;
  const double* bb = (s ? a : b) + (size_t)ch * Q;
  double* cc = c + (size_t)ch * Q;
  const int np = Q >> 1;  // Q even in this benchmark
  d2 va[NP], vb[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int j = (blockIdx.x * NP + k) * 256 + threadIdx.x;
    if (j < np) {
      if (NT) {
        va[k].x = __builtin_nontemporal_load(aa + 2 * j), va[k].y = __builtin_nontemporal_load(aa + 2 * j + 1);
        vb[k].x = __builtin_nontemporal_load(bb + 2 * j), vb[k].y = __builtin_nontemporal_load(bb + 2 * j + 1);
      } else {
        va[k] = *reinterpret_cast<const d2*>(aa + 2 * j);
        vb[k] = *reinterpret_cast<const d2*>(bb + 2 * j);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int j = (blockIdx.x * NP + k) * 256 + threadIdx.x;
    if (j < np) {
      d2 r;
      r.x = va[k].x - hh * vb[k].x, r.y = va[k].y - hh * vb[k].y;
      if (NT) {
        __builtin_nontemporal_store(r.x, cc + 2 * j);
        __builtin_nontemporal_store(r.y, cc + 2 * j + 1);
      } else {
        *reinterpret_cast<d2*>(cc + 2 * j) = r;
      }
    }
  }
}
// (3) persistent: a fixed number of workgroups strides over (chain, tile) pairs
template <bool NT>
__global__ void __launch_bounds__(256) k_persist(const double* a, const double* b, double* c, const int* ok, const int* cur,
                                                 const double* h, int Q, int B) {
  const int np = Q >> 1;
  const int tiles = (np + 511) / 512;  // 512 pairs per workgroup visit (two per work item)
  for (int w = blockIdx.x; w < tiles * B; w += gridDim.x) {
    const int ch = w / tiles, t = w - ch * tiles;
    if (!ok[ch]) continue;
    const int s = cur[ch];
    const double hh = h[ch];
    const double* aa = (s ? b : a) + (size_t)ch * Q;
    const double* bb = (s ? a : b) + (size_t)ch * Q;
    double* cc = c + (size_t)ch * Q;
    const int j0 = t * 512 + threadIdx.x, j1 = j0 + 256;
    d2 a0, b0, a1, b1;
    if (j0 < np) a0 = *reinterpret_cast<const d2*>(aa + 2 * j0), b0 = *reinterpret_cast<const d2*>(bb + 2 * j0);
    if (j1 < np) a1 = *reinterpret_cast<const d2*>(aa + 2 * j1), b1 = *reinterpret_cast<const d2*>(bb + 2 * j1);
    if (j0 < np) {
      d2 r;
      r.x = a0.x - hh * b0.x, r.y = a0.y - hh * b0.y;
      *reinterpret_cast<d2*>(cc + 2 * j0) = r;
    }
    if (j1 < np) {
      d2 r;
      r.x = a1.x - hh * b1.x, r.y = a1.y - hh * b1.y;
      *reinterpret_cast<d2*>(cc + 2 * j1) = r;
    }
  }
}
// (4) read-only column max of |a - b| (the reverse-check distance): two components per work item + wave reduction
template <int NP>
__global__ void __launch_bounds__(256) k_diffmax(const double* a, const double* b, unsigned long long* red, const int* ok,
                                                 int Q) {
  const int ch = blockIdx.y;
  if (!ok[ch]) return;
  const double* aa = a + (size_t)ch * Q;
  const double* bb = b + (size_t)ch * Q;
  const int np = Q >> 1;
  unsigned long long v = 0;
  d2 va[NP], vb[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int j = (blockIdx.x * NP + k) * 256 + threadIdx.x;
    if (j < np) va[k] = *reinterpret_cast<const d2*>(aa + 2 * j), vb[k] = *reinterpret_cast<const d2*>(bb + 2 * j);
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int j = (blockIdx.x * NP + k) * 256 + threadIdx.x;
    if (j < np) {
      const unsigned long long x = __double_as_longlong(fabs(va[k].x - vb[k].x)), y = __double_as_longlong(fabs(va[k].y - vb[k].y));
      v = x > v ? x : v;
      v = y > v ? y : v;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0 && v) atomicMax(red + ch, v);
}

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e = (x);                                            \
    if (e != hipSuccess) {                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e));                \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main() {
  const int B = 256, Q = 80106, NSET = 3, REP = 30;
  const size_t n = (size_t)B * Q;
  double *a[NSET], *b[NSET], *c[NSET];
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&a[i], n * 8 + 4096));
    CK(hipMalloc(&b[i], n * 8 + 4096));
    CK(hipMalloc(&c[i], n * 8 + 4096));
    CK(hipMemset(a[i], 0, n * 8));
    CK(hipMemset(b[i], 0, n * 8));
  }
  int *ok, *cur;
  double* h;
  unsigned long long* red;
  CK(hipMalloc(&ok, B * 4));
  CK(hipMalloc(&cur, B * 4));
  CK(hipMalloc(&h, B * 8));
  CK(hipMalloc(&red, B * 8));
  std::vector<int> one(B, 1), zero(B, 0);
  std::vector<double> hh(B, 0.1);
  CK(hipMemcpy(ok, one.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(cur, zero.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(h, hh.data(), B * 8, hipMemcpyHostToDevice));
  CK(hipMemset(red, 0, B * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int np = Q / 2;
  auto run = [&](const char* name, double bytes, auto launch) -> int {
    for (int i = 0; i < 3; ++i) launch(i % NSET);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < REP; ++i) launch(i % NSET);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms / REP * 1e3, bytes / (ms / REP * 1e-3) / 1e12);
    return 0;
  };
  const double tri = 3.0 * n * 8, rd2 = 2.0 * n * 8;
  run("flat1 (1 comp / item, tid / Q, flag loads)", tri, [&](int s) {
    hipLaunchKernelGGL(k_flat1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a[s], b[s], c[s], ok, cur, h, Q, (long)n);
  });
#define ROWS(NP, NT)                                                                                               \
  run("rows NP=" #NP " NT=" #NT, tri, [&](int s) {                                                                 \
    hipLaunchKernelGGL((k_rows<NP, NT>), dim3((np + 256 * NP - 1) / (256 * NP), B), dim3(256), 0, 0, a[s], b[s], c[s], ok, \
                       cur, h, Q);                                                                                 \
  })
  ROWS(1, false);
  ROWS(2, false);
  ROWS(4, false);
  ROWS(8, false);
  ROWS(1, true);
  ROWS(2, true);
  ROWS(4, true);
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "persistent, %d workgroups", g);
    run(nm, tri, [&](int s) { hipLaunchKernelGGL(k_persist<false>, dim3(g), dim3(256), 0, 0, a[s], b[s], c[s], ok, cur, h, Q, B); });
  }
  run("diffmax NP=1", rd2, [&](int s) { hipLaunchKernelGGL(k_diffmax<1>, dim3((np + 255) / 256, B), dim3(256), 0, 0, a[s], b[s], red, ok, Q); });
  run("diffmax NP=2", rd2, [&](int s) { hipLaunchKernelGGL(k_diffmax<2>, dim3((np + 511) / 512, B), dim3(256), 0, 0, a[s], b[s], red, ok, Q); });
  run("diffmax NP=4", rd2, [&](int s) { hipLaunchKernelGGL(k_diffmax<4>, dim3((np + 1023) / 1024, B), dim3(256), 0, 0, a[s], b[s], red, ok, Q); });
  // the same launches right behind a kernel that leaves 82 MB of dirty lines (as in the integrator's sequence)
  run("rows NP=2 after a writer (pair)", 2 * tri, [&](int s) {
    hipLaunchKernelGGL((k_rows<2, false>), dim3((np + 511) / 512, B), dim3(256), 0, 0, a[s], b[s], c[s], ok, cur, h, Q);
    hipLaunchKernelGGL((k_rows<2, false>), dim3((np + 511) / 512, B), dim3(256), 0, 0, c[s], b[(s + 1) % NSET], a[(s + 2) % NSET], ok, cur, h, Q);
  });
  // what a read-only kernel pays for the dirty lines its predecessors left in the caches: a writer of W MB, then the
  // reverse-check kernel alone between the events
  __attribute__((unused)) auto after_writer = [&](size_t wbytes) -> int {
    double* big;
    CK(hipMalloc(&big, wbytes));
    float tot = 0;
    for (int i = 0; i < 10; ++i) {
      CK(hipMemsetAsync(big, i, wbytes, 0));
      const int s = i % NSET;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_diffmax<1>, dim3((np + 255) / 256, B), dim3(256), 0, 0, a[s], b[s], red, ok, Q);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) tot += ms;
    }
    printf("diffmax NP=1 right after a %4zu MB fill        %7.1f us\n", wbytes >> 20, tot / 8 * 1e3);
    CK(hipFree(big));
    return 0;
  };
  after_writer((size_t)8 << 20);
  after_writer((size_t)82 << 20);
  after_writer((size_t)256 << 20);
  after_writer((size_t)656 << 20);
  after_writer((size_t)1300 << 20);
  return 0;
}
