// One instantiation of the steady-state sampler (denovo_lane_kernel.hpp) per object file (-DLANE_K=..), compiled in
// parallel with the other sampler objects.  The host API in mchap_hip.hip calls the entry points below; they are not
// part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_lane_kernel.hpp"

#define LANE_CAT_(a, k) a##k
#define LANE_CAT(a, k) LANE_CAT_(a, k)

extern "C" __attribute__((visibility("hidden"))) int LANE_CAT(mchap_lane_init_, LANE_K)(const double *ln, const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int LANE_CAT(mchap_lane_launch_, LANE_K)(const mchap::SimtParams *P, int lsh, unsigned grid,
                                                                                          size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_lane_kernel<LANE_K>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P, lsh);
  return (int)hipGetLastError();
}
