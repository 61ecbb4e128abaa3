// TEST-ONLY: host emulation build of the library's sequencing logic (see backend_emu.h).
#define CHMC_HD
#define CHMC_BACKEND_NAME "emu:host-TEST-ONLY"
#define CHMC_BACKEND_HEADER "backend_emu.h"
#include "../../manifold_mcmc_for_diffusions_amd/csrc/chmc_core.h"
#include "../../manifold_mcmc_for_diffusions_amd/csrc/chmc_api.inc"
