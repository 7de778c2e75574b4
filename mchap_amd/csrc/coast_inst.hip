// The coasting kernel of the phased sampler (denovo_coast_kernel.hpp) in its own object file.  The host API in
// mchap_hip.hip calls the entry point below; it is not part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_coast_kernel.hpp"

extern "C" __attribute__((visibility("hidden"))) int mchap_coast_launch(const mchap::SimtParams *P, unsigned grid, size_t lds,
                                                                        hipStream_t stream) {
  hipLaunchKernelGGL(mchap::denovo_coast_kernel, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}
