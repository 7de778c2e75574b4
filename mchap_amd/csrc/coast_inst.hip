// The coasting kernel of the phased sampler (denovo_coast_kernel.hpp) in its own object file.  The host API in
// mchap_hip.hip calls the entry point below; it is not part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_coast_kernel.hpp"

// wide != 0: a list of handed-back chains (few): COAST_NW_LIST wavefronts per chain, as many steps per sweep
extern "C" __attribute__((visibility("hidden"))) int mchap_coast_launch(const mchap::SimtParams *P, unsigned grid, int wide,
                                                                        hipStream_t stream) {
  if (wide)
    hipLaunchKernelGGL(mchap::denovo_coast_kernel<mchap::COAST_NW_LIST>, dim3(grid < 1024u ? grid : 1024u), dim3(64 * mchap::COAST_NW_LIST),
                       mchap::coast_lds_bytes(P->max_pos), stream, *P);
  else
    hipLaunchKernelGGL(mchap::denovo_coast_kernel<1>, dim3(grid), dim3(64), mchap::coast_lds_bytes(P->max_pos), stream, *P);
  return (int)hipGetLastError();
}
