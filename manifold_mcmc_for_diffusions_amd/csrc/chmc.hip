// libchmc_hip.so: HIP / gfx950 build of the batched constrained-HMC leapfrog library.
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o ../libchmc_hip.so chmc.hip
#include <hip/hip_runtime.h>
#define CHMC_HD __host__ __device__
#define CHMC_BACKEND_NAME "hip:gfx950"
#define CHMC_BACKEND_HEADER "backend_hip.h"
#define CHMC_WAVE_KERNELS 1
#include "chmc_core.h"
#include "chmc_wave.h"
#include "chmc_retract.h"
#include "chmc_api.inc"
