// One instantiation of the wave-per-chain sampler (kernel 1) per object file (-DV1_RPL=1/2/4/8/16); see spec_inst.hip.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_kernel.hpp"

#define INST_CAT_(a, k) a##k
#define INST_CAT(a, k) INST_CAT_(a, k)

extern "C" __attribute__((visibility("hidden"))) int INST_CAT(mchap_v1_init_, V1_RPL)(const double *ln, const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int INST_CAT(mchap_v1_launch_, V1_RPL)(const mchap::DenovoParams *P, unsigned gx,
                                                                                       unsigned gy, unsigned block, size_t lds,
                                                                                       hipStream_t stream) {
  auto kern = mchap::denovo_mcmc_kernel<V1_RPL>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(block), lds, stream, *P);
  return (int)hipGetLastError();
}
