// TEST-ONLY host "backend": runs every functor of chmc_core.h in a plain loop so that the kernel sequencing
// logic of chmc_api.inc can be unit-tested in the GPU-less build container.  NOT part of the product: the
// package loader (manifold_mcmc_for_diffusions_amd/_lib.py) only ever loads libchmc_hip.so.
#pragma once
#include <cstdlib>
#include <cstring>
static int dev_set(int) { return 0; }
static int dev_init(int) { return 0; }
static int dev_num_cus() { return 1; }
static void* dev_alloc(size_t bytes) { return calloc(bytes ? bytes : 8, 1); }
static void dev_free(void* p) { free(p); }
static void dev_zero(void* p, size_t bytes) { memset(p, 0, bytes); }
static void h2d(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); }
static void d2h(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); }
static void d2d(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
static int dev_sync() { return 0; }
static int g_poll_val[4];
static void use_stream(int) {}
static void streams_fork() {}
static void streams_join() {}
static void stagger_wait() {}
static void stagger_record() {}
static void poll_begin(int slot, const int* d) { g_poll_val[slot] = *d; }
static int poll_end(int slot) { return g_poll_val[slot]; }
template <class F>
static void launch(F f, long n, int cls = 0) {
  (void)cls;
  for (long t = 0; t < n; ++t) f((int)t);
}
// (HIP: the last workgroup of the check publishes the round's count into the pinned poll slot; here: read it directly)
template <class F>
static void launch_publish(F f, long n, int* counter, int poll_slot) {
  launch(f, n);
  g_poll_val[poll_slot] = *counter;
}
template <class F>
static void launch_rows(F f, int ncol, int B, int cls = 0) {
  (void)cls;
  for (int c = 0; c < B; ++c) {
    if (!f.active(c)) continue;
    for (int col = 0; col < ncol; col += 2) f(c, col);
  }
}
static int rowsum_groups(int ncol) { return ((ncol + 1) / 2 + 2047) / 2048; }  // 8 column pairs per work item x 256
template <class F>
static void launch_rowsum(F f, int ncol, int B, int nacc, double* partial, int cls = 0) {
  (void)cls;
  const int ng = rowsum_groups(ncol);
  for (int c = 0; c < B; ++c) {
    if (!f.active(c)) continue;
    for (int g = 0; g < ng; ++g) {
      double acc[CHMC_ROWSUM_MAX];
      for (int a = 0; a < CHMC_ROWSUM_MAX; ++a) acc[a] = 0.0;
      for (int col = g * 4096; col < ncol && col < (g + 1) * 4096; col += 2) f(c, col, acc);
      for (int a = 0; a < nacc; ++a) partial[((size_t)c * ng + g) * nacc + a] = acc[a];
    }
  }
}
template <class F>
static void launch_colmax(F f, int ncol, int B, int cls = 0) {
  (void)cls;
  for (int c = 0; c < B; ++c) {
    if (!f.active(c)) continue;
    unsigned long long v = 0ULL;
    for (int col = 0; col < ncol; ++col) {
      const unsigned long long o = f(c, col);
      v = o > v ? o : v;
    }
    unsigned long long* t = f.red(c);
    if (t && v > *t) *t = v;
    if constexpr (F::kFinish) {
      if (t && f.has_finish()) f.finish(c, *t);
    }
  }
}
extern "C" int chmc_profile_enable(int) { return 0; }
extern "C" int chmc_profile_stride(int) { return 0; }
extern "C" int chmc_profile_get(double* ms, long long* n) {
  for (int i = 0; i < 10; ++i) {
    if (ms) ms[i] = 0.0;
    if (n) n[i] = 0;
  }
  return 0;
}
