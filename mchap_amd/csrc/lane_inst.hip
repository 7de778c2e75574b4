// One ploidy of the steady-state pipeline (denovo_lane_kernel.hpp: settling kernel + steady kernel) per object file
// (-DLANE_K=..), compiled in parallel with the other sampler objects.  The host API in mchap_hip.hip calls the entry
// points below; they are not part of the C ABI.
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

// mode < 0: the steady kernel; else the settling kernel with that mode (LANE_MODE_*)
extern "C" __attribute__((visibility("hidden"))) int LANE_CAT(mchap_lane_launch_, LANE_K)(const mchap::SimtParams *P, int lsh, int mode,
                                                                                          unsigned grid, size_t lds, hipStream_t stream) {
  if (mode < 0) {
    hipLaunchKernelGGL(mchap::denovo_steady_kernel<LANE_K>, dim3(grid), dim3(64), lds, stream, *P, lsh);
    return (int)hipGetLastError();
  }
  auto ks = mchap::denovo_settle_kernel<LANE_K>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P, lsh, mode);
  return (int)hipGetLastError();
}

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
extern "C" __attribute__((visibility("hidden"))) int LANE_CAT(mchap_lane_stats_, LANE_K)(unsigned long long *out, int reset) {
  unsigned long long z[mchap::N_STATS] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mchap::g_stats), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(mchap::g_stats), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
