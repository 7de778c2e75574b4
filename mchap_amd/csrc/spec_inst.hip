// One instantiation of the speculative sampler per object file (-DSPEC_K=.. -DSPEC_G=..), so that the nine
// instantiations compile in parallel (each takes 20-60 s; in one translation unit the library took five minutes).
// The host API in mchap_hip.hip calls the two entry points below; they are not part of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/mchap_hip.h"
#include "denovo_spec_kernel.hpp"

#define SPEC_CAT_(a, k, g) a##k##_##g
#define SPEC_CAT(a, k, g) SPEC_CAT_(a, k, g)

#ifdef SPEC_PIPE
// the phased form (kernel 5) of this instantiation: its own object file, its own copy of the constant tables; with
// -DMCHAP_SPEC_VAR=1 / 2 the "side by side" / "deep" variants (denovo_spec_kernel.hpp; "specs" / "specd" objects)
#if MCHAP_SPEC_VAR == 1
#define mchap_specp_init_ mchap_specs_init_
#define mchap_specp_launch_ mchap_specs_launch_
#define mchap_specp_launchc_ mchap_specs_launchc_
#define mchap_specp_stats_ mchap_specs_stats_
#elif MCHAP_SPEC_VAR == 2
#define mchap_specp_init_ mchap_specd_init_
#define mchap_specp_launch_ mchap_specd_launch_
#define mchap_specp_launchc_ mchap_specd_launchc_
#define mchap_specp_stats_ mchap_specd_stats_
#endif
extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_specp_init_, SPEC_K, SPEC_G)(const double *ln,
                                                                                                 const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_specp_launch_, SPEC_K, SPEC_G)(
    const mchap::SimtParams *P, unsigned grid, size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_spec_kernel<SPEC_K, SPEC_G, true>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}

#if SPEC_G == 64
// ... with decision contexts per genotype (denovo_spec_kernel<.., CTX = true>): the launches over handed-back chains
extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_specp_launchc_, SPEC_K, SPEC_G)(
    const mchap::SimtParams *P, unsigned grid, size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_spec_kernel<SPEC_K, SPEC_G, true, MCHAP_SPEC_VAR, false, true>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}
#endif
#else
extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_spec_init_, SPEC_K, SPEC_G)(const double *ln,
                                                                                                const double *ln_inv) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln), ln, sizeof(double) * 260) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mchap::c_ln_inv), ln_inv, sizeof(double) * 260) != hipSuccess) return 1;
  return 0;
}

extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_spec_launch_, SPEC_K, SPEC_G)(
    const mchap::SimtParams *P, unsigned grid, size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_spec_kernel<SPEC_K, SPEC_G>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64), lds, stream, *P);
  return (int)hipGetLastError();
}

#if SPEC_G == 64
// one wavefront per replica of a temperature ladder (denovo_spec_kernel<.., TW = true>): a workgroup of n_temps wavefronts per chain
extern "C" __attribute__((visibility("hidden"))) int SPEC_CAT(mchap_spec_launchtw_, SPEC_K, SPEC_G)(
    const mchap::SimtParams *P, unsigned grid, size_t lds, hipStream_t stream) {
  auto ks = mchap::denovo_spec_kernel<SPEC_K, SPEC_G, false, MCHAP_SPEC_VAR, true>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ks, dim3(grid), dim3(64 * P->d.n_temps), lds, stream, *P);
  return (int)hipGetLastError();
}
#endif

#endif  // SPEC_PIPE

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
// profiling builds: this object's copy of the event counters
#ifdef SPEC_PIPE
#define SPEC_STATS_NAME SPEC_CAT(mchap_specp_stats_, SPEC_K, SPEC_G)
#else
#define SPEC_STATS_NAME SPEC_CAT(mchap_spec_stats_, SPEC_K, SPEC_G)
#endif
extern "C" __attribute__((visibility("hidden"))) int SPEC_STATS_NAME(unsigned long long *out, int reset) {
  unsigned long long z[mchap::N_STATS] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mchap::g_stats), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(mchap::g_stats), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

