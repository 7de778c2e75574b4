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

// log(2^e m) for a mantissa m in [0.5, 1) (or 0 / NaN / inf as v_frexp_mant returns them for such arguments); `x` only decides the
// special values: 0 -> -inf, negative or NaN -> NaN
__device__ __forceinline__ double read_log_core(double m, int e, double x) {
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

__device__ __forceinline__ double read_log(double x) {
  return read_log_core(__builtin_amdgcn_frexp_mant(x), __builtin_amdgcn_frexp_exp(x), x);
}

// Sum of the logarithms of N (<= 4) per-read terms as ONE logarithm (round 5): log(x_0 ... x_{N-1}) with the mantissas multiplied
// and the exponents summed, so that nothing underflows however small the terms are -- about 50 instructions instead of N x 40.
// Every likelihood kernel is bound by the logarithm (4 of them per lane and evaluation at configs[1]: 40 % of an evaluation's
// instructions; 45 of the 70 instructions per (genotype, read) term of the exact caller).  Used where the reads' weights are 0 / 1
// (no counts of de-duplicated rows): a read of weight 0 enters as the factor 1.  The value differs from the sum of the N logs in
// the last bits (three more roundings in the product, three fewer in the sum): every kernel that evaluates a likelihood takes
// the same groups of reads, so the kernels stay bit-identical with each other, and the parity tolerance against the oracle (1e-10
// relative) is five orders of magnitude above the difference.  A term of 0 gives -inf, a NaN gives NaN, as the sum would.
template <int N>
__device__ __forceinline__ double read_log_product(const double (&x)[N]) {
  static_assert(N >= 1 && N <= 4, "groups of at most four reads: the product of four mantissas stays above 2^-4");
  if constexpr (N == 1) return read_log(x[0]);
  double m = __builtin_amdgcn_frexp_mant(x[0]);
  int e = __builtin_amdgcn_frexp_exp(x[0]);
#pragma unroll
  for (int i = 1; i < N; i++) {
    m *= __builtin_amdgcn_frexp_mant(x[i]);
    e += __builtin_amdgcn_frexp_exp(x[i]);
  }
  // m in [2^-N, 1), or 0 (a term of 0), NaN, inf
  return read_log_core(__builtin_amdgcn_frexp_mant(m), e + __builtin_amdgcn_frexp_exp(m), m);
}
// The lane's partial sum over its N reads with weights w: one logarithm for the group where the unit's weights are 0 / 1
// (`grouped`, wave-uniform), else the weighted sum of the N logarithms in read order
template <int N>
__device__ __forceinline__ double read_log_sum(const double (&x)[N], const double (&w)[N], bool grouped) {
  if (grouped) {
    double y[N];
#pragma unroll
    for (int i = 0; i < N; i++) y[i] = w[i] != 0.0 ? x[i] : 1.0;
    return read_log_product<N>(y);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < N; i++) s += read_log(x[i]) * w[i];
  return s;
}

// ... over any number of chunks: groups of four chunks from chunk 0 on (what the samplers that take a unit's chunks in blocks of
// four form); without grouping the plain sum in chunk order
template <int N>
__device__ __forceinline__ double read_log_sum_chunks(const double (&x)[N], const double (&w)[N], bool grouped) {
  if constexpr (N <= 4) {
    return read_log_sum<N>(x, w, grouped);
  } else {
    static_assert(N % 4 == 0, "chunks in fours");
    double s = 0.0;
    if (grouped) {
#pragma unroll
      for (int g = 0; g < N / 4; g++) {
        double y[4];
#pragma unroll
        for (int i = 0; i < 4; i++) y[i] = w[4 * g + i] != 0.0 ? x[4 * g + i] : 1.0;
        s += read_log_product<4>(y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; i++) s += read_log(x[i]) * w[i];
    }
    return s;
  }
}

// test hook (mchap_read_log_batch)
static __global__ void read_log_kernel(const double *x, long long n, double *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = read_log(x[i]);
}

}  // namespace mchap
