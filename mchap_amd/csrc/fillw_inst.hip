// One ploidy of the table-completion kernel (denovo_fillw_kernel.hpp: workgroup per chain, wavefront per request) per object
// file (-DFILLW_K=..), compiled in parallel with the other sampler objects.  The host API in mchap_hip.hip calls the entry points
// below; they are not part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_fillw_kernel.hpp"

#define FILLW_CAT_(a, k) a##k
#define FILLW_CAT(a, k) FILLW_CAT_(a, k)

extern "C" __attribute__((visibility("hidden"))) int FILLW_CAT(mchap_fillw_init_, FILLW_K)(const double *ln, const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

// (lds: fillw_lds_bytes; wide != 0: genotypes of more than 64 bits, keyed by their changed words)
extern "C" __attribute__((visibility("hidden"))) int FILLW_CAT(mchap_fillw_launch_, FILLW_K)(const mchap::SimtParams *P, unsigned grid,
                                                                                            size_t lds, hipStream_t stream) {
  const bool wide = (P->fill_kw & 1) != 0, deep = (P->fill_kw & 2) != 0;
  auto ks = wide ? (deep ? mchap::denovo_fillw_kernel<FILLW_K, true, true> : mchap::denovo_fillw_kernel<FILLW_K, true, false>)
                 : (deep ? mchap::denovo_fillw_kernel<FILLW_K, false, true> : mchap::denovo_fillw_kernel<FILLW_K, false, false>);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(mchap::FILLW_NT), lds, stream, *P);
  return (int)hipGetLastError();
}
