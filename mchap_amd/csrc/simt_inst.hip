// One instantiation of the lanes-over-chains sampler per object file (-DSIMT_K=0/2/4/6/8); see spec_inst.hip.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_simt_kernel.hpp"

#define INST_CAT_(a, k) a##k
#define INST_CAT(a, k) INST_CAT_(a, k)
// -DSIMT_WIDE (with -DSIMT_K=0): haplotype words of 128 bits -- the general fallback for targets wider than 64 bits of sampled
// alleles per haplotype (object simt_w.o, entry points mchap_simt_init_w / mchap_simt_launch_w)
#ifdef SIMT_WIDE
#define SIMT_NAME w
#define SIMT_KERNEL mchap::denovo_simt_kernel<0, mchap::u128>
#else
#define SIMT_NAME SIMT_K
#define SIMT_KERNEL mchap::denovo_simt_kernel<SIMT_K>
#endif

extern "C" __attribute__((visibility("hidden"))) int INST_CAT(mchap_simt_init_, SIMT_NAME)(const double *ln, const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int INST_CAT(mchap_simt_launch_, SIMT_NAME)(const mchap::SimtParams *P, unsigned grid,
                                                                                         size_t lds, hipStream_t stream) {
  auto ks = SIMT_KERNEL;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}
