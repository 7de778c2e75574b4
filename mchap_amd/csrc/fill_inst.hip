// One ploidy of the table-completion kernel (denovo_fill_kernel.hpp) per object file (-DFILL_K=..), compiled in parallel with
// the other sampler objects.  The host API in mchap_hip.hip calls the entry points below; they are not part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_fill_kernel.hpp"

#define FILL_CAT_(a, k) a##k
#define FILL_CAT(a, k) FILL_CAT_(a, k)

extern "C" __attribute__((visibility("hidden"))) int FILL_CAT(mchap_fill_init_, FILL_K)(const double *ln, const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int FILL_CAT(mchap_fill_launch_, FILL_K)(const mchap::SimtParams *P, unsigned grid,
                                                                                          size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_fill_kernel<FILL_K>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
extern "C" __attribute__((visibility("hidden"))) int FILL_CAT(mchap_fill_stats_, FILL_K)(unsigned long long *out, int reset) {
  unsigned long long z[mchap::N_STATS] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mchap::g_stats), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(mchap::g_stats), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
