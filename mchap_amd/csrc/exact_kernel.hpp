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

// Ploidy bound of a kernel's genotype arrays and unrolled loops (template argument KM): 8 -- MCHAP_MAX_PLOIDY, what rounds 1-4
// took -- for ploidies up to 8, and EXACT_KMAX = 16 for ploidies 9 to 15 (round 5: the shapes the de novo sampler's general kernel
// assembles), so that the usual ploidies keep their code
constexpr int EXACT_KMAX = 16;
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
  int Rcap;                 // rows of the LDS product table (0: all R; less: the passes tile the reads, exact_tile)
  double *ptab_ext;         // product table + read weights of THIS workgroup in global memory instead of LDS, or null
  double ptab_scale;        // the table holds product * ptab_scale: 1 / ploidy for the exact caller -- the factor every term of a read's sum
                            // over a genotype's haplotypes carries (likelihood.py:60-66), applied once per table entry instead of once per
                            // (genotype, read, haplotype): the same products, a third of the arithmetic less -- 1 for the call sampler (its
                            // greedy start divides by the partial ploidies)
  // outputs / workspace
  float *llk32;             // [U][G] or null
  double *llk64;            // [U][G] or null
  double *ljoint;           // [U][G] or null (workspace for posterior_mode)
  double *part_max;         // [U][nblk] block maxima of ljoint
  long long *part_idx;      // [U][nblk] index of the block maximum (first occurrence)
  double *part_llk;         // [U][nblk] llk at the block maximum
  double *part_lse;         // [U][nblk] log-sum-exp of the block's ljoint
  // second pass of the streaming form (exact_pass2_kernel): inputs written by exact_mode_kernel, per-block partial sums out
  const double *unit_total;     // [U] log normaliser
  const int64_t *unit_mode;     // [U][K] mode alleles
  double *part_freq;            // [U][nblk][2H + 1]: allele counts, allele occurrence, mode-support probability
};

// C(n + k - 1, k), the number of genotypes of ploidy k over n alleles (jitutils.py:228-250; 0 for n == 0)
__device__ __forceinline__ long long cwr(int n, int k) {
  if (n <= 0) return 0;
  long long r = 1;
  for (int d = 1; d <= k; d++) r = r * (n - 1 + d) / d;
  return r;
}

// jitutils.py:279-318: VCF index -> ascending alleles
template <int KM>
__device__ __forceinline__ void unrank_genotype(long long index, int K, int (&g)[KM]) {
  long long remainder = index;
#pragma unroll
  for (int p = KM; p >= 1; p--) {
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

// LDS of the enumeration passes: product table, read weights, prior tables, then the caller's scratch
struct ExactLds {
  double *ptab;   // [R][H]
  double *cnt;    // [R]
  double *lgd;    // [H][K+1]
  double *lgf;    // [K+1]
  double *lfreq;  // [H]
  double *red;    // reduction scratch (the rest of the allocation)
  bool w01;       // every read weight of the unit is 1: ONE logarithm per four consecutive reads (read_log_product, round 5)
};
// Builds P[r][h], the read weights and the prior tables of `unit` (whole workgroup; ends with a barrier).
// (with P.Rcap < R only the first Rcap reads are tabulated here: exact_tile brings in the others, tile by tile)
__device__ __forceinline__ void exact_setup(const ExactParams &P, int unit, unsigned char *smem, ExactLds &E, PriorTab &pt) {
  const int M = P.M, A = P.A, H = P.H, K = P.K;
  const int R = (P.Rcap > 0 && P.Rcap < P.R) ? P.Rcap : P.R;
  if (P.ptab_ext) {  // (call sampler with a table that exceeds the LDS: slower reads, no limit)
    E.ptab = P.ptab_ext;
    E.cnt = E.ptab + (size_t)R * H;
    E.lgd = reinterpret_cast<double *>(smem);
  } else {
    E.ptab = reinterpret_cast<double *>(smem);
    E.cnt = E.ptab + (size_t)R * H;
    E.lgd = E.cnt + R;
  }
  E.lgf = E.lgd + (size_t)H * (K + 1);
  E.lfreq = E.lgf + (K + 1);
  E.red = E.lfreq + H;
  __shared__ double s_left;
  const double *reads = P.reads + (size_t)unit * P.R * M * A;
  const int8_t *haps = P.haps + (size_t)unit * H * M;
  for (int q = threadIdx.x; q < R * H; q += blockDim.x) {
    const int r = q / H, h = q % H;
    double prod = 1.0;
    for (int j = 0; j < M; j++) {
      const double v = reads[((size_t)r * M + j) * A + haps[h * M + j]];
      if (!isnan(v)) prod *= v;  // assemble/likelihood.py:54-59
    }
    E.ptab[q] = prod * P.ptab_scale;
  }
  for (int r = threadIdx.x; r < R; r += blockDim.x) E.cnt[r] = P.counts ? (double)P.counts[(size_t)unit * P.R + r] : 1.0;
  {
    // (over ALL reads of the unit, not only the first tile's: the groups of four run through the tiles)
    int weighted = 0;
    if (P.counts)
      for (int r = threadIdx.x; r < P.R; r += blockDim.x) weighted |= (P.counts[(size_t)unit * P.R + r] != 1) ? 1 : 0;
    E.w01 = __syncthreads_or(weighted) == 0;
  }
  double F = 0.0;
  const bool has_prior = P.has_prior != 0;
  const bool has_freqs = has_prior && P.freqs != nullptr;
  if (has_prior) {
    F = P.inbreeding[unit];
    const double scale = (1.0 - F) / F;
    for (int q = threadIdx.x; q < H * (K + 1); q += blockDim.x) {
      const int h = q / (K + 1), d = q % (K + 1);
      const double alpha = has_freqs ? P.freqs[(size_t)unit * H + h] * scale : (1.0 / (double)H) * scale;
      E.lgd[q] = (F == 0.0 || d == 0) ? 0.0 : lgamma((double)d + alpha) - (lgamma((double)d + 1.0) + lgamma(alpha));
    }
    for (int d = threadIdx.x; d <= K; d += blockDim.x) E.lgf[d] = lgamma((double)d + 1.0);
    for (int h = threadIdx.x; h < H; h += blockDim.x) E.lfreq[h] = has_freqs ? P.freqs[(size_t)unit * H + h] : 0.0;
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
  pt.lgd = E.lgd;
  pt.lgf = E.lgf;
  pt.lfreq = E.lfreq;
  pt.left = has_prior ? s_left : 0.0;
  pt.lnH = log((double)H);
  pt.F = F;
  pt.has_freqs = has_freqs ? 1 : 0;
}

// The exact kernels' product table and read weights are always in the LDS (only the call sampler's exact_setup may be
// handed a table in global memory), but the pointer in ExactLds is generic and the compiler then reads it with flat_load
// (through the vector memory path: r03 PMC, 6.4 VMEM reads per wavefront-term and no LDS instructions).  Reading through
// an LDS-qualified pointer gives ds_read_b64: same values, shorter latency, and the vector memory unit stays free.
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ lds_cdouble *lds_table(const double *p) { return (lds_cdouble *)p; }

// log likelihood of the genotype with ascending alleles g (calling/exact.py:252-263 via assemble/likelihood.py:17-70)
template <int KM>
__device__ __forceinline__ double exact_llk(const ExactLds &E, const int (&g)[KM], int R, int H, int K, double invK) {
  double llk = 0.0;
  lds_cdouble *ptab = lds_table(E.ptab), *cnt = lds_table(E.cnt);
  int r = 0;
  if (E.w01) {  // unweighted reads: the sum over four consecutive reads as the logarithm of their product
    for (; r + 4 <= R; r += 4) {
      double rp[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        lds_cdouble *row = ptab + (r + t) * H;
        rp[t] = 0.0;
#pragma unroll
        for (int k = 0; k < KM; k++)
          if (k < K) rp[t] += row[g[k]];
      }
      llk += read_log_product<4>(rp);
    }
  }
  for (; r < R; r++) {
    lds_cdouble *row = ptab + r * H;
    double rp = 0.0;
#pragma unroll
    for (int k = 0; k < KM; k++)
      if (k < K) rp += row[g[k]];
    llk += read_log(rp) * cnt[r];
  }
  return llk;
}

// NQ genotypes of one thread at once: each sum runs over the reads in order as in exact_llk (same values); the NQ
// chains of LDS reads, adds and logs are independent and hide each other's latency (two waves per SIMD is all the
// 64 KB table leaves).  MI355X, config #4: pass 1 28.0 -> 21.8 (two) -> 20.4 ms (four at a time).
template <int NQ, int KM>
__device__ __forceinline__ void exact_llkn(const ExactLds &E, const int (&g)[NQ][KM], int R, int H, int K, double invK,
                                           double (&l)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; q++) l[q] = 0.0;
  lds_cdouble *ptab = lds_table(E.ptab), *cnt = lds_table(E.cnt);
  int r = 0;
  if (E.w01) {  // (as exact_llk: groups of four consecutive reads, one logarithm each)
    for (; r + 4 <= R; r += 4) {
      double rp[NQ][4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        lds_cdouble *row = ptab + (r + t) * H;
#pragma unroll
        for (int q = 0; q < NQ; q++) rp[q][t] = 0.0;
#pragma unroll
        for (int k = 0; k < KM; k++)
          if (k < K) {
#pragma unroll
            for (int q = 0; q < NQ; q++) rp[q][t] += row[g[q][k]];
          }
      }
#pragma unroll
      for (int q = 0; q < NQ; q++) l[q] += read_log_product<4>(rp[q]);
    }
  }
  for (; r < R; r++) {
    lds_cdouble *row = ptab + r * H;
    double rp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) rp[q] = 0.0;
#pragma unroll
    for (int k = 0; k < KM; k++)
      if (k < K) {
#pragma unroll
        for (int q = 0; q < NQ; q++) rp[q] += row[g[q][k]];
      }
    const double w = cnt[r];
#pragma unroll
    for (int q = 0; q < NQ; q++) l[q] += read_log(rp[q]) * w;
  }
}

// Reads r0 .. r0 + rn - 1 of `unit` into the product table and the weights (whole workgroup; barriers on both sides)
__device__ __forceinline__ void exact_tile(const ExactParams &P, int unit, const ExactLds &E, int r0, int rn) {
  const int M = P.M, A = P.A, H = P.H;
  const double *reads = P.reads + ((size_t)unit * P.R + r0) * M * A;
  const int8_t *haps = P.haps + (size_t)unit * H * M;
  __syncthreads();
  for (int q = threadIdx.x; q < rn * H; q += blockDim.x) {
    const int r = q / H, h = q % H;
    double prod = 1.0;
    for (int j = 0; j < M; j++) {
      const double v = reads[((size_t)r * M + j) * A + haps[h * M + j]];
      if (!isnan(v)) prod *= v;
    }
    E.ptab[q] = prod * P.ptab_scale;
  }
  for (int r = threadIdx.x; r < rn; r += blockDim.x) E.cnt[r] = P.counts ? (double)P.counts[(size_t)unit * P.R + r0 + r] : 1.0;
  __syncthreads();
}
// The log likelihoods of the thread's genotypes lo + threadIdx.x + t * blockDim.x (t < NGT) when the reads do not fit
// the LDS at once: tile by tile, every genotype's sum continued in read order -- the values of the untiled loop.
constexpr int EXACT_NGT = EXACT_GENOS_PER_BLOCK / EXACT_THREADS;
template <int KM>
__device__ __forceinline__ void exact_llk_tiled(const ExactParams &P, int unit, const ExactLds &E, long long lo, long long hi,
                                                double (&llk)[EXACT_NGT]) {
  const int H = P.H, K = P.K, cap = P.Rcap;
  const double invK = 1.0 / (double)K;
#pragma unroll
  for (int t = 0; t < EXACT_NGT; t++) llk[t] = 0.0;
  for (int r0 = 0; r0 < P.R; r0 += cap) {
    const int rn = min(cap, P.R - r0);
    if (r0 > 0) exact_tile(P, unit, E, r0, rn);  // (exact_setup brought the first tile)
#pragma unroll
    for (int t = 0; t < EXACT_NGT; t++) {
      const long long i = lo + threadIdx.x + (long long)t * EXACT_THREADS;
      if (i < hi) {
        int g[KM];
        unrank_genotype(i, K, g);
        double acc = llk[t];
        lds_cdouble *ptab = lds_table(E.ptab), *cnt = lds_table(E.cnt);
        int r = 0;
        // (the host keeps the tile a multiple of four reads, so the groups of four are those of the untiled loop)
        if (E.w01) {
          for (; r + 4 <= rn; r += 4) {
            double rp[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              lds_cdouble *row = ptab + (r + q) * H;
              rp[q] = 0.0;
#pragma unroll
              for (int k = 0; k < KM; k++)
                if (k < K) rp[q] += row[g[k]];
            }
            acc += read_log_product<4>(rp);
          }
        }
        for (; r < rn; r++) {
          lds_cdouble *row = ptab + r * H;
          double rp = 0.0;
#pragma unroll
          for (int k = 0; k < KM; k++)
            if (k < K) rp += row[g[k]];
          acc += read_log(rp) * cnt[r];
        }
        llk[t] = acc;
      }
    }
  }
}

template <bool TILED, int KM = 8>
__global__ __launch_bounds__(EXACT_THREADS) void exact_pass1_kernel(const ExactParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int unit = blockIdx.y;
  const int R = P.R, H = P.H, K = P.K;
  const bool has_prior = P.has_prior != 0;
  ExactLds E;
  PriorTab pt;
  exact_setup(P, unit, smem, E, pt);
  double *red = E.red;  // [5][EXACT_THREADS] reduction scratch

  const long long G = P.G;
  const long long lo = (long long)blockIdx.x * EXACT_GENOS_PER_BLOCK;
  long long hi = lo + EXACT_GENOS_PER_BLOCK;
  if (hi > G) hi = G;
  const double invK = 1.0 / (double)K;
  double best = -INFINITY, best_llk = -INFINITY;
  long long best_idx = 0x7fffffffffffffffll;
  double lse_m = -INFINITY, lse_s = 0.0;  // running log-sum-exp: max and scaled sum
  auto visit = [&](long long i, const int (&g)[KM], double llk) {
    const size_t o = (size_t)unit * G + i;
    if (P.llk32) P.llk32[o] = (float)llk;  // calling/exact.py:254 float32 store
    if (P.llk64) P.llk64[o] = llk;
    if (P.ljoint || P.part_max) {
      const double lpr = has_prior ? calling_log_prior(pt, g, K) : 0.0;
      const double lj = llk + lpr;
      if (P.ljoint) P.ljoint[o] = lj;
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
  };
  if constexpr (TILED) {
    double tl[EXACT_NGT];
    exact_llk_tiled<KM>(P, unit, E, lo, hi, tl);
#pragma unroll
    for (int t = 0; t < EXACT_NGT; t++) {
      const long long i = lo + threadIdx.x + (long long)t * EXACT_THREADS;
      if (i < hi) {
        int g[KM];
        unrank_genotype(i, K, g);
        visit(i, g, tl[t]);
      }
    }
  } else {
    // the thread's genotypes i, i + 256, ... two at a time (visited in index order: the first maximum stays the first)
    long long i = lo + threadIdx.x;
    for (; i + 3 * (long long)blockDim.x < hi; i += 4 * (long long)blockDim.x) {
      int g4[4][KM];
      double l4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) unrank_genotype(i + (long long)q * blockDim.x, K, g4[q]);
      exact_llkn<4>(E, g4, R, H, K, invK, l4);
#pragma unroll
      for (int q = 0; q < 4; q++) visit(i + (long long)q * blockDim.x, g4[q], l4[q]);
    }
    for (; i + (long long)blockDim.x < hi; i += 2 * (long long)blockDim.x) {
      int g2[2][KM];
      double l2[2];
      unrank_genotype(i, K, g2[0]);
      unrank_genotype(i + blockDim.x, K, g2[1]);
      exact_llkn<2>(E, g2, R, H, K, invK, l2);
      visit(i, g2[0], l2[0]);
      visit(i + blockDim.x, g2[1], l2[1]);
    }
    if (i < hi) {
      int g[KM];
      unrank_genotype(i, K, g);
      visit(i, g, exact_llk(E, g, R, H, K, invK));
    }
  }
  if (!P.part_max) return;
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

// ---- streaming form without any per-genotype array (calling/exact.py:156-249: two passes over the genotypes) ----
// After pass 1: normaliser and mode of every unit from the per-block partials.
struct ExactModeParams {
  ExactParams e;
  int64_t *mode_alleles;                  // [U][K] (output, and input of the second pass)
  double *mode_llk, *mode_prob, *total;   // [U]
};
__global__ __launch_bounds__(64) void exact_mode_kernel(const ExactModeParams P) {
  constexpr int KM = EXACT_KMAX;
  const ExactParams &E = P.e;
  const int unit = blockIdx.x;
  if (threadIdx.x != 0) return;
  double b = -INFINITY, bl = -INFINITY, total = -INFINITY;
  long long bi = 0;
  for (int q = 0; q < E.nblk; q++) {
    const size_t o = (size_t)unit * E.nblk + q;
    if (E.part_max[o] > b) {  // blocks are in index order: strict > keeps the first maximum (calling/exact.py:51)
      b = E.part_max[o];
      bi = E.part_idx[o];
      bl = E.part_llk[o];
    }
    total = add_log_prob(total, E.part_lse[o]);
  }
  int g[KM];
  unrank_genotype(bi, E.K, g);
  for (int k = 0; k < E.K; k++) P.mode_alleles[(size_t)unit * E.K + k] = g[k];
  if (P.mode_llk) P.mode_llk[unit] = bl;
  if (P.mode_prob) P.mode_prob[unit] = exp(b - total);
  P.total[unit] = total;
}

// Second pass: every genotype's joint value again, its probability under the known normaliser added to the allele
// count / occurrence sums (calling/exact.py:108-153) and, when it consists of exactly the mode's alleles, to the
// support probability (64-105).  A wavefront takes 64 consecutive genotypes at a time; each of the 2H + 1 sums gets
// the wavefront's 64 contributions by a shuffle reduction (fixed order) and is kept per wavefront in LDS -- a
// kilobyte instead of a column per thread, so that two workgroups share a CU as in pass 1.  Wavefronts are summed
// in order, blocks in order by exact_freq_kernel: deterministic.
inline size_t exact_pass2_lds(int R, int H, int K, int threads) {
  return ((size_t)R * H + R + (size_t)H * (K + 1) + (K + 1) + H + (size_t)(2 * H + 1) * (threads / 64)) * 8;
}
// HAVE_LJ: pass 1 left llk + log prior of every genotype in P.ljoint (the caller's workspace had room for U * G
// doubles): the pass then reads them back instead of forming every likelihood a second time -- same values, same sums.
template <bool TILED, bool HAVE_LJ = false, int KM = 8>
__global__ __launch_bounds__(EXACT_THREADS) void exact_pass2_kernel(const ExactParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int unit = blockIdx.y;
  const int R = P.R, H = P.H, K = P.K;
  const bool has_prior = !HAVE_LJ && P.has_prior != 0;
  ExactLds E;
  PriorTab pt;
  if constexpr (HAVE_LJ) {
    E.red = reinterpret_cast<double *>(smem);  // only the per-wavefront sums live in LDS
  } else {
    exact_setup(P, unit, smem, E, pt);
  }
  const int nt = blockDim.x, nw = nt >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int NS = 2 * H + 1;
  double *acc = E.red;  // [nw][2H + 1]
  for (int i = threadIdx.x; i < nw * NS; i += nt) acc[i] = 0.0;
  __syncthreads();
  const double total = P.unit_total[unit];
  int ms[KM], ns = 0;  // distinct alleles of the mode, ascending
  for (int k = 0; k < K; k++) {
    const int a = (int)P.unit_mode[(size_t)unit * K + k];
    if (k == 0 || a != ms[ns - 1]) ms[ns++] = a;
  }
  const long long G = P.G;
  const long long lo = (long long)blockIdx.x * EXACT_GENOS_PER_BLOCK;
  long long hi = lo + EXACT_GENOS_PER_BLOCK;
  if (hi > G) hi = G;
  const double invK = 1.0 / (double)K;
  double *mine = acc + (size_t)wave * NS;
  double tl[EXACT_NGT];
  if constexpr (HAVE_LJ) {
#pragma unroll
    for (int t = 0; t < EXACT_NGT; t++) {
      const long long i = lo + threadIdx.x + (long long)t * nt;
      tl[t] = i < hi ? P.ljoint[(size_t)unit * G + i] : 0.0;
    }
  } else if constexpr (TILED) {
    exact_llk_tiled<KM>(P, unit, E, lo, hi, tl);  // (nt == EXACT_THREADS: thread t's genotypes lo + t + 256 q)
  } else {
    // the likelihoods of the thread's genotypes, four at a time (exact_llkn), ahead of the accounting loop
#pragma unroll
    for (int tqq = 0; tqq < EXACT_NGT; tqq += 4) {
      const long long ia = lo + threadIdx.x + (long long)tqq * nt;
#pragma unroll
      for (int q = 0; q < 4; q++) tl[tqq + q] = 0.0;
      if (ia + 3 * (long long)nt < hi) {
        int g4[4][KM];
        double l4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) unrank_genotype(ia + (long long)q * nt, K, g4[q]);
        exact_llkn<4>(E, g4, R, H, K, invK, l4);
#pragma unroll
        for (int q = 0; q < 4; q++) tl[tqq + q] = l4[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (ia + (long long)q * nt < hi) {
            int ga[KM];
            unrank_genotype(ia + (long long)q * nt, K, ga);
            tl[tqq + q] = exact_llk(E, ga, R, H, K, invK);
          }
        }
      }
    }
  }
#pragma unroll
  for (int tqq = 0; tqq < EXACT_NGT; tqq++) {
    const long long i0 = lo + (long long)wave * 64 + (long long)tqq * nt;  // wave-uniform
    if (i0 >= hi) continue;
    const long long i = i0 + lane;
    int g[KM];
#pragma unroll
    for (int k = 0; k < KM; k++) g[k] = -1;
    double prob = 0.0;
    bool support = false;
    if (i < hi) {
      unrank_genotype(i, K, g);
      const double llk = tl[tqq];
      const double lpr = has_prior ? calling_log_prior(pt, g, K) : 0.0;
      prob = exp((llk + lpr) - total);
      int nd = 0;
      bool same = true;
      for (int k = 0; k < K; k++) {
        if (k == 0 || g[k] != g[k - 1]) {
          same = same && nd < ns && ms[nd] == g[k];
          nd++;
        }
      }
      support = same && nd == ns;
    }
    for (int a = 0; a < H; a++) {
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < KM; k++) cnt += (k < K && g[k] == a) ? 1 : 0;
      const double c1 = wave_sum(prob * (double)cnt);         // allele count
      const double c2 = wave_sum(cnt > 0 ? prob : 0.0);       // allele occurrence
      if (lane == 0) {
        mine[a] += c1;
        mine[H + a] += c2;
      }
    }
    const double c3 = wave_sum(support ? prob : 0.0);
    if (lane == 0) mine[2 * H] += c3;
  }
  __syncthreads();
  for (int h = threadIdx.x; h < NS; h += nt) {
    double sum = 0.0;
    for (int w = 0; w < nw; w++) sum += acc[(size_t)w * NS + h];
    P.part_freq[((size_t)unit * P.nblk + blockIdx.x) * NS + h] = sum;
  }
}
struct ExactFreqParams {
  const double *part_freq;  // [U][nblk][2H + 1]
  int nblk, H, K;
  double *support_prob;     // [U] or null
  double *freqs_out, *occur_out;  // [U][H] or null
};
__global__ __launch_bounds__(64) void exact_freq_kernel(const ExactFreqParams P) {
  const int unit = blockIdx.x;
  for (int h = threadIdx.x; h < 2 * P.H + 1; h += blockDim.x) {
    double sum = 0.0;
    for (int q = 0; q < P.nblk; q++) sum += P.part_freq[((size_t)unit * P.nblk + q) * (2 * P.H + 1) + h];
    if (h < P.H) {
      if (P.freqs_out) P.freqs_out[(size_t)unit * P.H + h] = sum / (double)P.K;
    } else if (h < 2 * P.H) {
      if (P.occur_out) P.occur_out[(size_t)unit * P.H + (h - P.H)] = sum;
    } else if (P.support_prob) {
      P.support_prob[unit] = sum;
    }
  }
}

// ---- array form for a batch: genotype_posteriors on stored likelihoods (calling/exact.py:295-329), one workgroup per unit,
// the joint values parked in the output array itself, and what call_exact.py:126-159 / exact.py:332-407 derive from the
// posterior array: its first maximum, the probability of the mode's support, allele frequencies / counts / occurrence ----
struct ExactArrayParams {
  const float *llk32;       // [U][G] or null
  const double *llk64;      // [U][G] or null (used when llk32 is null)
  double *post;             // [U][G] out (may be null when only... never: the summaries read it)
  const double *post_in;    // [U][G] summaries of a given posterior array instead (llk32 == llk64 == null)
  long long G;
  int K, H, has_prior;
  const double *inbreeding; // [U]
  const double *freqs;      // [U][H] or null
  int64_t *mode_alleles;    // [U][K] or null
  double *mode_prob, *support_prob;       // [U] or null
  double *afreq, *acount, *aoccur;        // [U][H] or null
  int nacc;                 // threads that accumulate the allele sums (an LDS column each)
};
constexpr int EXACT_ARRAY_THREADS = 1024;
inline size_t exact_array_lds(int H, int K, int nacc) {
  return ((size_t)H * (K + 1) + (K + 1) + H + 2 * (size_t)EXACT_ARRAY_THREADS + (size_t)(2 * H + 1) * nacc) * 8;
}
template <int KM = 8>
__global__ __launch_bounds__(EXACT_ARRAY_THREADS) void exact_array_kernel(const ExactArrayParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int unit = blockIdx.x;
  const int K = P.K, H = P.H;
  const long long G = P.G;
  double *lgd = reinterpret_cast<double *>(smem);
  double *lgf = lgd + (size_t)H * (K + 1);
  double *lfreq = lgf + (K + 1);
  double *red = lfreq + H;                    // [2][threads]
  double *acc = red + 2 * EXACT_ARRAY_THREADS;  // [2H + 1][nacc]
  __shared__ double s_left, s_total;
  __shared__ long long s_mode;
  const int nt = blockDim.x;
  const double *post;
  if (P.post_in) {
    post = P.post_in + (size_t)unit * G;
  } else {
    // ---- genotype_posteriors ----
    const bool is_f32 = P.llk32 != nullptr;
    const bool has_freqs = P.has_prior && P.freqs;
    const double F = P.has_prior ? P.inbreeding[unit] : 0.0;
    if (P.has_prior) {
      const double scale = (1.0 - F) / F;
      for (int q = threadIdx.x; q < H * (K + 1); q += nt) {
        const int h = q / (K + 1), d = q % (K + 1);
        const double alpha = has_freqs ? P.freqs[(size_t)unit * H + h] * scale : (1.0 / (double)H) * scale;
        lgd[q] = (F == 0.0 || d == 0) ? 0.0 : lgamma((double)d + alpha) - (lgamma((double)d + 1.0) + lgamma(alpha));
      }
      for (int d = threadIdx.x; d <= K; d += nt) lgf[d] = lgamma((double)d + 1.0);
      for (int h = threadIdx.x; h < H; h += nt) lfreq[h] = has_freqs ? P.freqs[(size_t)unit * H + h] : 0.0;
      if (threadIdx.x == 0) {
        double sum_alpha = 0.0;
        if (has_freqs) {
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
    pt.left = P.has_prior ? s_left : 0.0;
    pt.lnH = log((double)H);
    pt.F = F;
    pt.has_freqs = has_freqs ? 1 : 0;
    double *out = P.post + (size_t)unit * G;
    double m = -INFINITY, sacc = 0.0;
    for (long long i = threadIdx.x; i < G; i += nt) {
      int g[KM];
      unrank_genotype(i, K, g);
      const double lpr = P.has_prior ? calling_log_prior(pt, g, K) : 0.0;
      double j;
      if (is_f32) j = (double)(float)((double)P.llk32[(size_t)unit * G + i] + lpr);  // stored in the array's dtype (exact.py:317)
      else j = P.llk64[(size_t)unit * G + i] + lpr;
      out[i] = j;
      if (j > m) {
        sacc = sacc * exp(m - j) + 1.0;
        m = j;
      } else if (j > -INFINITY) {
        sacc += exp(j - m);
      }
    }
    red[threadIdx.x] = m;
    red[EXACT_ARRAY_THREADS + threadIdx.x] = sacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double mm = -INFINITY, ss = 0.0;
      for (int t = 0; t < nt; t++) {
        if (red[t] > -INFINITY) {
          if (red[t] > mm) {
            ss = ss * exp(mm - red[t]) + red[EXACT_ARRAY_THREADS + t];
            mm = red[t];
          } else {
            ss += red[EXACT_ARRAY_THREADS + t] * exp(red[t] - mm);
          }
        }
      }
      // (float64 also for float32 likelihoods: the compiled reference's log-denominator is float64 -- numba unifies
      // add_log_prob's result type -- and only the joint values are stored in float32; oracle: orc_genotype_posteriors_f32)
      s_total = mm + log(ss);
    }
    __syncthreads();
    const double total = s_total;
    for (long long i = threadIdx.x; i < G; i += nt) out[i] = exp(out[i] - total);
    __syncthreads();
    post = out;
  }
  const bool want_mode = P.mode_alleles || P.mode_prob || P.support_prob;
  const bool want_freq = P.afreq || P.acount || P.aoccur;
  if (!want_mode && !want_freq) return;
  // ---- first maximum of the posterior array (np.argmax) ----
  {
    double b = -INFINITY;
    long long bi = 0x7fffffffffffffffll;
    for (long long i = threadIdx.x; i < G; i += nt) {
      const double v = post[i];
      if (v > b) {
        b = v;
        bi = i;
      }
    }
    red[threadIdx.x] = b;
    red[EXACT_ARRAY_THREADS + threadIdx.x] = (double)bi;
    __syncthreads();
    if (threadIdx.x == 0) {
      double bb = -INFINITY, bidx = 9.0e18;
      for (int t = 0; t < nt; t++)
        if (red[t] > bb || (red[t] == bb && red[EXACT_ARRAY_THREADS + t] < bidx)) {
          bb = red[t];
          bidx = red[EXACT_ARRAY_THREADS + t];
        }
      s_mode = bidx < 9.0e18 ? (long long)bidx : 0;
      if (P.mode_prob) P.mode_prob[unit] = bb;
    }
    __syncthreads();
  }
  int mg[KM];
  unrank_genotype(s_mode, K, mg);
  if (P.mode_alleles && threadIdx.x < K) P.mode_alleles[(size_t)unit * K + threadIdx.x] = mg[threadIdx.x];
  int ms[KM], ns = 0;
  for (int k = 0; k < K; k++)
    if (k == 0 || mg[k] != mg[k - 1]) ms[ns++] = mg[k];
  // ---- sums over the genotypes: P.nacc accumulating threads (an LDS column each), summed in thread order ----
  const int na = P.nacc;
  if ((int)threadIdx.x < na) {
    for (int h = 0; h < 2 * H + 1; h++) acc[(size_t)h * na + threadIdx.x] = 0.0;
    for (long long i = threadIdx.x; i < G; i += na) {
      int g[KM];
      unrank_genotype(i, K, g);
      const double prob = post[i];
      int nd = 0;
      bool same = true;
      for (int k = 0; k < K; k++) {
        acc[(size_t)g[k] * na + threadIdx.x] += prob;
        if (k == 0 || g[k] != g[k - 1]) {
          acc[(size_t)(H + g[k]) * na + threadIdx.x] += prob;
          same = same && nd < ns && ms[nd] == g[k];
          nd++;
        }
      }
      if (same && nd == ns) acc[(size_t)(2 * H) * na + threadIdx.x] += prob;
    }
  }
  __syncthreads();
  for (int h = threadIdx.x; h < 2 * H + 1; h += nt) {
    double sum = 0.0;
    for (int t = 0; t < na; t++) sum += acc[(size_t)h * na + t];
    if (h < H) {
      if (P.afreq) P.afreq[(size_t)unit * H + h] = sum / (double)K;
      if (P.acount) P.acount[(size_t)unit * H + h] = sum;
    } else if (h < 2 * H) {
      if (P.aoccur) P.aoccur[(size_t)unit * H + (h - H)] = sum;
    } else if (P.support_prob) {
      P.support_prob[unit] = sum;
    }
  }
}

}  // namespace mchap
