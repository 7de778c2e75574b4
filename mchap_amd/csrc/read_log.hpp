// Natural logarithm for the per-read terms of the likelihoods (gfx950).
//
// Every likelihood in MCHap is sum_r count_r * log(p_r) (assemble/likelihood.py:54-59, calling/likelihood.py), and
// the kernels that evaluate it are bound by the logarithm: the device library's log() is ~100 VALU instructions
// (double-double arithmetic).  read_log() is the classic argument reduction x = 2^k (1 + f), sqrt(1/2) <= 1 + f <
// sqrt(2), with s = f / (2 + f) and a degree-14 odd polynomial in s (the coefficients of Sun's fdlibm e_log.c, a
// Remez fit on [0, 0.1716]): ~35 instructions, error < 1 ulp (0.86 ulp over 2e7 inputs against long double on the
// host; it differs from a correctly rounded log by one ulp for ~4 % of arguments).  The parity tolerances on
// log-likelihoods (1e-10 relative) are seven orders of magnitude above that.
//   x == 0 -> -inf, x < 0 or NaN -> NaN, denormals handled by v_frexp.
#pragma once
#include <hip/hip_runtime.h>

namespace mchap {

__device__ __forceinline__ double read_log(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const bool small = m < 0.70710678118654752440;
  m = small ? m * 2.0 : m;
  e = small ? e - 1 : e;
  const double f = m - 1.0;
  const double d = 2.0 + f;  // [1.707, 2.414)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-d, s, f), r, s);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double k = (double)e;
  double v = k * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + k * 1.90821492927058770002e-10)) - f);
  v = x == 0.0 ? -INFINITY : v;
  v = x > 0.0 || x == 0.0 ? v : NAN;
  return v;
}

// test hook (mchap_read_log_batch)
static __global__ void read_log_kernel(const double *x, long long n, double *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = read_log(x[i]);
}

}  // namespace mchap
