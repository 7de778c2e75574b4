// Exact genotype caller for MI355X (gfx950): enumeration of all C(H+K-1, K) genotypes over H known
// haplotypes (reference calling/exact.py:17-369, calling/prior.py:116-179, calling/utils.py:7-35,
// jitutils.py:114-146,254-318).
//
// Per unit: the per-(read, haplotype) products P[r][h] = prod_j reads[r, j, hap[h][j]] (NaN skipped) are built
// once per workgroup in LDS, row r holding H consecutive doubles, so a wavefront whose lanes each own one
// genotype reads P[r][g_k] for a common r from one H*8-byte window: distinct haplotypes fall in distinct banks
// (H <= 32) and equal ones broadcast.  Each lane then runs the reference's own sequential sum over reads for
// its genotype; genotypes are visited in VCF order (index -> alleles by combinatorial unranking).
#pragma once
#include "denovo_kernel.hpp"

namespace mchap {

constexpr int EXACT_THREADS = 256;
constexpr int EXACT_GENOS_PER_BLOCK = 4096;

struct ExactParams {
  const double *reads;      // [U][R][M][A]
  const int64_t *counts;    // [U][R] or null
  const int8_t *haps;       // [U][H][M]
  const double *inbreeding; // [U] or null (no prior)
  const double *freqs;      // [U][H] or null
  int R, M, A, H, K;
  long long G;
  int has_prior;
  int nblk;
  // outputs / workspace
  float *llk32;             // [U][G] or null
  double *llk64;            // [U][G] or null
  double *ljoint;           // [U][G] or null (workspace for posterior_mode)
  double *part_max;         // [U][nblk] block maxima of ljoint
  long long *part_idx;      // [U][nblk] index of the block maximum (first occurrence)
  double *part_llk;         // [U][nblk] llk at the block maximum
  double *part_lse;         // [U][nblk] log-sum-exp of the block's ljoint
};

// C(n + k - 1, k), the number of genotypes of ploidy k over n alleles (jitutils.py:228-250; 0 for n == 0)
__device__ __forceinline__ long long cwr(int n, int k) {
  if (n <= 0) return 0;
  long long r = 1;
  for (int d = 1; d <= k; d++) r = r * (n - 1 + d) / d;
  return r;
}

// jitutils.py:279-318: VCF index -> ascending alleles
__device__ __forceinline__ void unrank_genotype(long long index, int K, int (&g)[MCHAP_MAX_PLOIDY]) {
  long long remainder = index;
#pragma unroll
  for (int p = MCHAP_MAX_PLOIDY; p >= 1; p--) {
    if (p > K) {
      g[p - 1] = 0;
      continue;
    }
    int n = -1;
    long long nw = 0, prev = 0;
    while (nw <= remainder) {
      n += 1;
      prev = nw;
      nw = cwr(n, p);
    }
    n -= 1;
    remainder -= prev;
    g[p - 1] = n;
  }
}

// jitutils.py:253-276
__device__ inline long long rank_genotype(const int *g, int K) {
  long long idx = 0;
  for (int i = 0; i < K; i++) idx += cwr(g[i], i + 1);
  return idx;
}

// jitutils.py:114-146
__device__ inline void increment_genotype(int *g, int K) {
  if (K == 1) {
    g[0] += 1;
    return;
  }
  const int previous = g[0];
  for (int i = 1; i < K; i++) {
    if (g[i] == previous) continue;
    g[i - 1] += 1;
    for (int z = 0; z < i - 1; z++) g[z] = 0;
    return;
  }
  g[K - 1] += 1;
  for (int z = 0; z < K - 1; z++) g[z] = 0;
}

// prior tables in LDS: lgd[h][d] = lgamma(d + alpha_h) - (lgamma(d + 1) + lgamma(alpha_h)), d = 0..K
struct PriorTab {
  const double *lgd;   // [H][K+1]
  const double *lgf;   // [K+1] lgamma(d + 1)
  const double *lfreq; // [H] prior allele frequencies (F == 0 with frequencies)
  double left;         // lgamma(K+1) + lgamma(sum_alpha) - lgamma(K + sum_alpha)
  double lnH;          // log(H)
  double F;
  int has_freqs;
};

// calling/prior.py:116-179 on ascending alleles g[0..K-1]
__device__ inline double calling_log_prior(const PriorTab &t, const int *g, int K) {
  // allelic dosage in first-occurrence order (calling/utils.py:7-35); g is ascending so runs are contiguous
  if (t.F == 0.0) {
    double den = 0.0;
    int i = 0;
    double prod = 1.0;
    // dosage array holds the run length at the first copy and 0 elsewhere: lgamma(0 + 1) = 0 for the zeros
    while (i < K) {
      int j = i;
      while (j < K && g[j] == g[i]) j++;
      den += t.lgf[j - i];
      i = j;
    }
    const double ln_perms = t.lgf[K] - den;
    if (!t.has_freqs) return ln_perms - (double)K * t.lnH;
    for (int q = 0; q < K; q++) prod *= t.lfreq[g[q]];
    return ln_perms + log(prod);
  }
  double prod = 0.0;
  int i = 0;
  while (i < K) {
    int j = i;
    while (j < K && g[j] == g[i]) j++;
    prod += t.lgd[g[i] * (K + 1) + (j - i)];
    i = j;
  }
  return t.left + prod;
}

__global__ __launch_bounds__(EXACT_THREADS) void exact_pass1_kernel(const ExactParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int unit = blockIdx.y;
  const int R = P.R, M = P.M, A = P.A, H = P.H, K = P.K;
  double *ptab = reinterpret_cast<double *>(smem);            // [R][H]
  double *cnt = ptab + (size_t)R * H;                         // [R]
  double *lgd = cnt + R;                                      // [H][K+1]
  double *lgf = lgd + (size_t)H * (K + 1);                    // [K+1]
  double *lfreq = lgf + (K + 1);                              // [H]
  double *red = lfreq + H;                                    // [5][EXACT_THREADS] reduction scratch
  __shared__ double s_left;

  const double *reads = P.reads + (size_t)unit * R * M * A;
  const int8_t *haps = P.haps + (size_t)unit * H * M;
  for (int q = threadIdx.x; q < R * H; q += blockDim.x) {
    const int r = q / H, h = q % H;
    double prod = 1.0;
    for (int j = 0; j < M; j++) {
      const double v = reads[((size_t)r * M + j) * A + haps[h * M + j]];
      if (!isnan(v)) prod *= v;  // assemble/likelihood.py:54-59
    }
    ptab[q] = prod;
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) cnt[r] = P.counts ? (double)P.counts[(size_t)unit * R + r] : 1.0;
  double F = 0.0;
  const bool has_prior = P.has_prior != 0;
  const bool has_freqs = has_prior && P.freqs != nullptr;
  if (has_prior) {
    F = P.inbreeding[unit];
    const double scale = (1.0 - F) / F;
    for (int q = threadIdx.x; q < H * (K + 1); q += blockDim.x) {
      const int h = q / (K + 1), d = q % (K + 1);
      const double alpha = has_freqs ? P.freqs[(size_t)unit * H + h] * scale : (1.0 / (double)H) * scale;
      lgd[q] = (F == 0.0 || d == 0) ? 0.0 : lgamma((double)d + alpha) - (lgamma((double)d + 1.0) + lgamma(alpha));
    }
    for (int d = threadIdx.x; d <= K; d += blockDim.x) lgf[d] = lgamma((double)d + 1.0);
    for (int h = threadIdx.x; h < H; h += blockDim.x) lfreq[h] = has_freqs ? P.freqs[(size_t)unit * H + h] : 0.0;
    if (threadIdx.x == 0) {
      double sum_alpha;
      if (has_freqs) {
        sum_alpha = 0.0;
        for (int h = 0; h < H; h++) sum_alpha += P.freqs[(size_t)unit * H + h] * scale;
      } else {
        sum_alpha = ((1.0 / (double)H) * scale) * (double)H;
      }
      s_left = (F == 0.0) ? 0.0 : (lgamma((double)K + 1.0) + lgamma(sum_alpha)) - lgamma((double)K + sum_alpha);
    }
  }
  __syncthreads();
  PriorTab pt;
  pt.lgd = lgd;
  pt.lgf = lgf;
  pt.lfreq = lfreq;
  pt.left = has_prior ? s_left : 0.0;
  pt.lnH = log((double)H);
  pt.F = F;
  pt.has_freqs = has_freqs ? 1 : 0;

  const long long G = P.G;
  const long long lo = (long long)blockIdx.x * EXACT_GENOS_PER_BLOCK;
  long long hi = lo + EXACT_GENOS_PER_BLOCK;
  if (hi > G) hi = G;
  const double invK = 1.0 / (double)K;
  double best = -INFINITY, best_llk = -INFINITY;
  long long best_idx = 0x7fffffffffffffffll;
  double lse_m = -INFINITY, lse_s = 0.0;  // running log-sum-exp: max and scaled sum
  for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int g[MCHAP_MAX_PLOIDY];
    unrank_genotype(i, K, g);
    double llk = 0.0;
    for (int r = 0; r < R; r++) {
      const double *row = ptab + (size_t)r * H;
      double rp = 0.0;
#pragma unroll
      for (int k = 0; k < MCHAP_MAX_PLOIDY; k++)
        if (k < K) rp += row[g[k]] * invK;
      llk += log(rp) * cnt[r];
    }
    const size_t o = (size_t)unit * G + i;
    if (P.llk32) P.llk32[o] = (float)llk;  // calling/exact.py:254 float32 store
    if (P.llk64) P.llk64[o] = llk;
    if (P.ljoint) {
      const double lpr = has_prior ? calling_log_prior(pt, g, K) : 0.0;
      const double lj = llk + lpr;
      P.ljoint[o] = lj;
      if (lj > best) {  // calling/exact.py:51: strict, first maximum wins
        best = lj;
        best_idx = i;
        best_llk = llk;
      }
      if (lj > lse_m) {
        lse_s = lse_s * exp(lse_m - lj) + 1.0;
        lse_m = lj;
      } else if (lj > -INFINITY) {
        lse_s += exp(lj - lse_m);
      }
    }
  }
  if (!P.ljoint) return;
  // block reduction
  double *rb = red, *ri = red + EXACT_THREADS, *rl = red + 2 * EXACT_THREADS, *rm = red + 3 * EXACT_THREADS, *rs = red + 4 * EXACT_THREADS;
  rb[threadIdx.x] = best;
  ri[threadIdx.x] = (double)best_idx;  // indices < 2^53
  rl[threadIdx.x] = best_llk;
  rm[threadIdx.x] = lse_m;
  rs[threadIdx.x] = lse_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = -INFINITY, bl = -INFINITY, bi = 9.0e18, m = -INFINITY, s = 0.0;
    for (int t = 0; t < (int)blockDim.x; t++) {
      if (rb[t] > b || (rb[t] == b && ri[t] < bi)) {
        b = rb[t];
        bi = ri[t];
        bl = rl[t];
      }
      if (rm[t] > -INFINITY) {
        if (rm[t] > m) {
          s = s * exp(m - rm[t]) + rs[t];
          m = rm[t];
        } else {
          s += rs[t] * exp(rm[t] - m);
        }
      }
    }
    const size_t o = (size_t)unit * P.nblk + blockIdx.x;
    P.part_max[o] = b;
    P.part_idx[o] = (long long)bi;
    P.part_llk[o] = bl;
    P.part_lse[o] = (m > -INFINITY) ? m + log(s) : -INFINITY;
  }
}

struct ExactFinalParams {
  ExactParams e;
  int64_t *mode_alleles;  // [U][K]
  double *mode_llk, *mode_prob, *support_prob;  // [U]
  double *freqs_out, *occur_out;                // [U][H] or null
};

// calling/exact.py:156-249 after the enumeration: normaliser, mode, support probability (64-105), and the
// posterior allele frequency / occurrence pass (108-153) over the stored joint values.
__global__ __launch_bounds__(256) void exact_finalize_kernel(const ExactFinalParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const ExactParams &E = P.e;
  const int unit = blockIdx.x;
  const int K = E.K, H = E.H;
  const long long G = E.G;
  __shared__ double s_total;
  double *acc = reinterpret_cast<double *>(smem);  // [2][H][blockDim]
  if (threadIdx.x == 0) {
    double b = -INFINITY, bl = -INFINITY, total = -INFINITY;
    long long bi = 0;
    for (int q = 0; q < E.nblk; q++) {
      const size_t o = (size_t)unit * E.nblk + q;
      if (E.part_max[o] > b) {  // blocks are in index order: strict > keeps the first maximum
        b = E.part_max[o];
        bi = E.part_idx[o];
        bl = E.part_llk[o];
      }
      total = add_log_prob(total, E.part_lse[o]);
    }
    s_total = total;
    int g[MCHAP_MAX_PLOIDY];
    unrank_genotype(bi, K, g);
    for (int k = 0; k < K; k++) P.mode_alleles[(size_t)unit * K + k] = g[k];
    P.mode_llk[unit] = bl;
    P.mode_prob[unit] = exp(b - total);
    if (P.support_prob) {
      // calling/exact.py:64-105: every dosage variant of the mode's support, itertools order
      int support[MCHAP_MAX_PLOIDY], ns = 0;
      for (int k = 0; k < K; k++)
        if (k == 0 || g[k] != g[k - 1]) support[ns++] = g[k];
      const int rem = K - ns;
      int idx[MCHAP_MAX_PLOIDY];
      for (int k = 0; k < rem; k++) idx[k] = 0;
      double slj = -INFINITY;
      bool more = true;
      while (more) {
        int tmp[MCHAP_MAX_PLOIDY];
        for (int k = 0; k < ns; k++) tmp[k] = support[k];
        for (int k = 0; k < rem; k++) tmp[ns + k] = support[idx[k]];
        for (int a = 1; a < K; a++) {  // insertion sort
          const int v = tmp[a];
          int b2 = a - 1;
          while (b2 >= 0 && tmp[b2] > v) {
            tmp[b2 + 1] = tmp[b2];
            b2--;
          }
          tmp[b2 + 1] = v;
        }
        slj = add_log_prob(slj, E.ljoint[(size_t)unit * G + rank_genotype(tmp, K)]);
        more = false;
        if (rem > 0) {
          int q = rem - 1;
          while (q >= 0 && idx[q] == ns - 1) q--;
          if (q >= 0) {
            const int v = idx[q] + 1;
            for (int z = q; z < rem; z++) idx[z] = v;
            more = true;
          }
        }
      }
      P.support_prob[unit] = exp(slj - total);
    }
  }
  __syncthreads();
  if (!P.freqs_out && !P.occur_out) return;
  const double total = s_total;
  const int nt = blockDim.x;
  for (int h = 0; h < H; h++) {
    acc[(size_t)h * nt + threadIdx.x] = 0.0;
    acc[(size_t)(H + h) * nt + threadIdx.x] = 0.0;
  }
  for (long long i = threadIdx.x; i < G; i += nt) {
    int g[MCHAP_MAX_PLOIDY];
    unrank_genotype(i, K, g);
    const double prob = exp(E.ljoint[(size_t)unit * G + i] - total);
    for (int k = 0; k < K; k++) {
      acc[(size_t)g[k] * nt + threadIdx.x] += prob;
      if (k == 0 || g[k] != g[k - 1]) acc[(size_t)(H + g[k]) * nt + threadIdx.x] += prob;
    }
  }
  __syncthreads();
  for (int h = threadIdx.x; h < 2 * H; h += nt) {
    double s = 0.0;
    for (int t = 0; t < nt; t++) s += acc[(size_t)h * nt + t];
    if (h < H) {
      if (P.freqs_out) P.freqs_out[(size_t)unit * H + h] = s / (double)K;
    } else if (P.occur_out) {
      P.occur_out[(size_t)unit * H + (h - H)] = s;
    }
  }
}

// calling/exact.py:295-329 on a stored likelihood array: priors, joint values stored in the array's dtype
// (float32 when is_f32), the float32-typed log-sum-exp of jitutils.py:7-74, float64 result array.
struct ExactPostParams {
  const float *llk32;
  const double *llk64;
  double *out;
  long long G;
  int K, H, has_prior, is_f32;
  double F;
  const double *freqs;
  double *scratch;  // [G] joint values
};

__global__ __launch_bounds__(1024) void exact_posteriors_kernel(const ExactPostParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int K = P.K, H = P.H;
  double *lgd = reinterpret_cast<double *>(smem);
  double *lgf = lgd + (size_t)H * (K + 1);
  double *lfreq = lgf + (K + 1);
  double *red = lfreq + H;  // [2][1024]
  __shared__ double s_left, s_total;
  const bool has_freqs = P.has_prior && P.freqs;
  const double F = P.F;
  if (P.has_prior) {
    const double scale = (1.0 - F) / F;
    for (int q = threadIdx.x; q < H * (K + 1); q += blockDim.x) {
      const int h = q / (K + 1), d = q % (K + 1);
      const double alpha = has_freqs ? P.freqs[h] * scale : (1.0 / (double)H) * scale;
      lgd[q] = (F == 0.0 || d == 0) ? 0.0 : lgamma((double)d + alpha) - (lgamma((double)d + 1.0) + lgamma(alpha));
    }
    for (int d = threadIdx.x; d <= K; d += blockDim.x) lgf[d] = lgamma((double)d + 1.0);
    for (int h = threadIdx.x; h < H; h += blockDim.x) lfreq[h] = has_freqs ? P.freqs[h] : 0.0;
    if (threadIdx.x == 0) {
      double sum_alpha = 0.0;
      if (has_freqs) {
        for (int h = 0; h < H; h++) sum_alpha += P.freqs[h] * scale;
      } else {
        sum_alpha = ((1.0 / (double)H) * scale) * (double)H;
      }
      s_left = (F == 0.0) ? 0.0 : (lgamma((double)K + 1.0) + lgamma(sum_alpha)) - lgamma((double)K + sum_alpha);
    }
  }
  __syncthreads();
  PriorTab pt;
  pt.lgd = lgd;
  pt.lgf = lgf;
  pt.lfreq = lfreq;
  pt.left = P.has_prior ? s_left : 0.0;
  pt.lnH = log((double)H);
  pt.F = F;
  pt.has_freqs = has_freqs ? 1 : 0;
  double m = -INFINITY, s = 0.0;
  for (long long i = threadIdx.x; i < P.G; i += blockDim.x) {
    int g[MCHAP_MAX_PLOIDY];
    unrank_genotype(i, K, g);
    const double lpr = P.has_prior ? calling_log_prior(pt, g, K) : 0.0;
    double j;
    if (P.is_f32) j = (double)(float)((double)P.llk32[i] + lpr);
    else j = P.llk64[i] + lpr;
    P.scratch[i] = j;
    if (j > m) {
      s = s * exp(m - j) + 1.0;
      m = j;
    } else if (j > -INFINITY) {
      s += exp(j - m);
    }
  }
  red[threadIdx.x] = m;
  red[1024 + threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mm = -INFINITY, ss = 0.0;
    for (int t = 0; t < (int)blockDim.x; t++) {
      if (red[t] > -INFINITY) {
        if (red[t] > mm) {
          ss = ss * exp(mm - red[t]) + red[1024 + t];
          mm = red[t];
        } else {
          ss += red[1024 + t] * exp(red[t] - mm);
        }
      }
    }
    double total = mm + log(ss);
    if (P.is_f32) total = (double)(float)total;  // the reference's accumulator is float32 here
    s_total = total;
  }
  __syncthreads();
  const double total = s_total;
  for (long long i = threadIdx.x; i < P.G; i += blockDim.x) {
    if (P.is_f32) P.out[i] = (double)expf((float)P.scratch[i] - (float)total);
    else P.out[i] = exp(P.scratch[i] - total);
  }
}

}  // namespace mchap
