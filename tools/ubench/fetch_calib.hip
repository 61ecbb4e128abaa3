// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library.
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane coalesced streaming read; "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads
// a known number of bytes (buffers far larger than the 256 MiB Infinity Cache, each byte read once) with one of the
// patterns the library's kernel classes use; tools/fetch_calib.py runs this binary under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv
// and stores bytes_read / (FETCH_SIZE * 1024) per pattern in profiles/fetch_calibration.json, from which
// tools/pmc_summary.py takes the factor of each kernel class.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib.bin fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2_t __attribute__((ext_vector_type(2)));

// (a) wide coalesced: 16 B per lane, consecutive lanes consecutive addresses (element-wise kernels, KUpdatePB)
__global__ void __launch_bounds__(256) calib_coalesced_16B(const d2_t* a, double* sink, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const d2_t v = a[i];
  if (v.x + v.y == 1.2345e301) sink[0] = v.x;
}
// (b) 8 B per lane coalesced: a wavefront reads 512 contiguous bytes per instruction (the wave-scan kernels'
//     per-step loads: k_newton_ivl, k_newton_lean, k_gld_*, k_jw_pb with X V = 4 read 8 B per lane and operand)
__global__ void __launch_bounds__(256) calib_coalesced_8B(const double* a, double* sink, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = a[i];
  if (v == 1.2345e301) sink[0] = v;
}
// (c) lane-per-stream: every lane walks its own contiguous stream with 16-B loads, 128 bytes (one 8-step tile of noise
//     increments) at a time; the 64 lanes of a wavefront are `stride` bytes apart (k_fwd_scan: one lane per block)
__global__ void __launch_bounds__(64) calib_lane_stream_16B(const char* a, double* sink, long stride, int tiles) {
  const long lane = (long)blockIdx.x * 64 + threadIdx.x;
  const char* p = a + lane * stride;
  double acc = 0.0;
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const d2_t v = *reinterpret_cast<const d2_t*>(p + (long)t * 128 + k * 16);
      acc += v.x + v.y;
    }
  }
  if (acc == 1.2345e301) sink[0] = acc;
}
// (d) 32 B per lane coalesced (two 16-B loads of consecutive addresses per lane: PB rows, X V = 4 doubles per step)
__global__ void __launch_bounds__(256) calib_coalesced_32B(const d2_t* a, double* sink, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const d2_t v = a[2 * i], u = a[2 * i + 1];
  if (v.x + v.y + u.x + u.y == 1.2345e301) sink[0] = v.x;
}

int main() {
  const long bytes = 2L << 30;  // 2 GiB per pattern, 8x the Infinity Cache
  char* buf = nullptr;
  double* sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
    fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    const long n16 = bytes / 16, n8 = bytes / 8, n32 = bytes / 32;
    hipLaunchKernelGGL(calib_coalesced_16B, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const d2_t*)buf, sink, n16);
    hipLaunchKernelGGL(calib_coalesced_8B, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, 0, (const double*)buf, sink, n8);
    hipLaunchKernelGGL(calib_coalesced_32B, dim3((unsigned)((n32 + 255) / 256)), dim3(256), 0, 0, (const d2_t*)buf, sink, n32);
    // 5120 lanes (configs[1]: 256 chains x 20 blocks), 32 000 bytes of noise increments per block of 2000 steps:
    // stride = one block's stream; repeated over the buffer so that 2 GiB are read in total by 13 launches of 160 MiB
    const long stride = 2000L * 16, lanes = 5120;
    const int tiles = (int)(stride / 128);
    const long per_launch = lanes * stride;
    for (long off = 0; off + per_launch <= bytes; off += per_launch)
      hipLaunchKernelGGL(calib_lane_stream_16B, dim3((unsigned)(lanes / 64)), dim3(64), 0, 0, buf + off, sink, stride, tiles);
    hipDeviceSynchronize();
  }
  printf("bytes_per_launch coalesced_16B %ld coalesced_8B %ld coalesced_32B %ld lane_stream_16B %ld\n", bytes, bytes, bytes,
         5120L * 2000L * 16);
  hipFree(buf);
  hipFree(sink);
  return 0;
}
