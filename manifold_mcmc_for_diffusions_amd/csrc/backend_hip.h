// HIP backend primitives used by chmc_api.inc: one non-blocking stream per process, every kernel is the
// generic 1-D launcher below instantiated with one functor of chmc_core.h.
#pragma once
#include <hip/hip_runtime.h>

static hipStream_t g_stream = nullptr;
static int g_device = -1;
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
  if (!g_stream || g_device != device) {
    e = hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      g_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
      return -1;
    }
    g_device = device;
  }
  g_first_err = hipSuccess;
  return 0;
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
static void d2d(void* dst, const void* src, size_t bytes) {
  note(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
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
template <class F>
static void launch(F f, long n) {
  if (n <= 0) return;
  const int bs = n >= 65536 ? 256 : 64;
  const unsigned grid = (unsigned)((n + bs - 1) / bs);
  hipLaunchKernelGGL(k_run<F>, dim3(grid), dim3(bs), 0, g_stream, f, (int)n);
  note(hipGetLastError());
}
// HIP events on the library's stream (used by bench.py to time kernels where they are launched)
extern "C" void* chmc_stream(void) { return (void*)g_stream; }
