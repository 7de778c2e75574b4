// De-novo MCMC haplotype assembler for MI355X (gfx950): one (locus x sample) unit per
// workgroup, one wavefront per chain, lanes over reads.
//
// Reference path (all paths relative to /root/reference/mchap/):
//   DenovoMCMC.fit/_mcmc            assemble/mcmc.py:103-265
//   _denovo_assembler               assemble/mcmc.py:268-426
//   mutation.base_step/compound     assemble/mutation.py:14-246
//   structural.*                    assemble/structural.py:22-673
//   log_likelihood*                 assemble/likelihood.py:17-148
//   prior.log_genotype_prior        assemble/prior.py:15-112
//   tempering.chain_swap_step       assemble/tempering.py:10-151
//   snp_posterior / homozygosity    assemble/snpcalling.py:14-70, assemble/mcmc.py:494-541
//   _read_mean_dist / sample        assemble/mcmc.py:455-491, jitutils.py:464-498
//
// Data layout.  The unit's float64 read tensor [R][M0][A] is staged once into LDS
// transposed to [M0*A][RPAD] (read index fastest) with NaN (gap) entries replaced by 1.0,
// the multiplicative identity the reference's NaN-skip amounts to (likelihood.py:56-59);
// rows r >= R are 1.0 with count 0.  Lane l owns reads l, l+64, ... so every LDS row read
// is a conflict-free 512-byte sweep.  A haplotype over the sampled (non-fixed) positions is
// one packed uint64 (position 0 in the most significant field), so haplotype equality,
// segment labels and the canonical sort are integer compares.  All chain state is
// wave-private; after the staging barrier the chains of a unit never synchronise.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mchap_hip.h"
#include "philox.hpp"

namespace mchap {

constexpr int WAVE = 64;
constexpr uint32_t SLOT_INIT = 0xFFFFu;

// log(n), log(1/n) for small n, computed by the host's libm at library load.
static __constant__ double c_ln[260];      // per translation unit (the library is built from several)
static __constant__ double c_ln_inv[260];

struct DenovoParams {
  const mchap_unit *units;
  const double *reads;
  const int64_t *counts;
  const int8_t *n_alleles;
  const int8_t *initial;
  uint64_t *trace;
  double *llks;
  int8_t *fixed;
  int32_t *status;
  const double *break_table;  // device copy, [(max_pos+1)][max_pos]
  int max_pos;
  int steps, chains, n_temps, n_intervals;
  double temps[MCHAP_MAX_TEMPS];
  double fix_hom, p_recomb, p_partial, p_dosage;
  uint64_t seed;
  int rpad;  // 64 * RPL
  // optional per-chain likelihood cache in HBM/L2: [units*chains][cache_slots] of {tag, llk}; 0 slots = off
  uint64_t *cache;
  int cache_slots;  // power of two
  // genotypes wider than 63 bits are tagged by a hash; their full words [units*chains][cache_slots][cache_key_words]
  // are kept beside the entries and compared on every tag match, so a hit is always the genotype itself
  // (cache_key_words = the batch's largest ploidy; 0 when every unit's genotypes fit the tag)
  uint64_t *cache_keys;
  int cache_key_words;
  // compact input (kernels 2 / 3 only): int8 allele calls [R][M0] per unit instead of the float64 tensor, optional
  // int16 base qualities, and the probability of a correct call per quality (qual_prob[0] when there are none)
  const int8_t *calls;
  const int16_t *quals;
  const double *qual_prob;
  int qual_prob_len;
};

// ---- wave-private LDS scratch -------------------------------------------------------------
struct WaveLayout {
  int w, pw, llk, rngn, sub, shift, hetrow, nal, probs, llks, optin, prior, buf, total;
};

__host__ __device__ inline int align8(int x) { return (x + 7) & ~7; }

__host__ __device__ inline int snv_genotypes(int n_alleles, int ploidy) {
  // C(n + k - 1, k)
  long r = 1;
  for (int d = 1; d <= ploidy; d++) r = r * (n_alleles - 1 + d) / d;
  return (int)r;
}

__host__ __device__ inline WaveLayout wave_layout(int K, int M, int A, int T) {
  WaveLayout L;
  int o = 0;
  L.w = o; o += 8 * T * K;
  L.pw = o; o += 8 * K;
  L.llk = o; o += 8 * T;
  L.rngn = o; o += 8 * (T + 1);
  L.probs = o; o += 8 * (K * K + 2 > A ? K * K + 2 : A);
  L.llks = o; o += 8 * (K * K + 2 > A ? K * K + 2 : A);
  L.prior = o; o += 8 * (2 * (K + 1) + 4);
  int nb = snv_genotypes(A, K);
  if (nb < M * A) nb = M * A;
  if (nb < M + 4) nb = M + 4;  // also holds the interval end points of a structural step (ints)
  L.buf = o; o += 8 * nb;
  L.optin = o; o += 4 * K * K;
  L.sub = o; o += align8(2 * K * M);
  L.hetrow = o; o += align8(2 * M);
  L.shift = o; o += align8(M);
  L.nal = o; o += align8(M);
  L.total = align8(o);
  return L;
}

__host__ __device__ inline int allele_bits(int A) { return A <= 2 ? 1 : (A <= 4 ? 2 : 3); }

// ---- small device helpers ------------------------------------------------------------------
// Sum over the 64 lanes, in every lane, associated as the XOR butterfly 32, 16, 8, 4, 2, 1 associates it (every likelihood of
// every kernel goes through this tree: the bits of a value must not depend on which kernel formed it).  The two steps across rows of
// 16 lanes are ds_bpermute round trips; inside a row the partners come through DPP row rotations (round 4: ~10 instead of ~100
// cycles of latency per step).  row_ror:8 reads lane (i + 8) mod 16 = i ^ 8 of the row; after that step the row's values have period
// 8, so row_ror:4 reads a lane holding exactly the value of lane i ^ 4 -- and so on: the same operands in every addition (a + b
// is b + a bit for bit), hence the same bits as the plain butterfly (tests/test_gpu_read_log.py::test_wave_sum_tree).
template <int CTRL>
__device__ __forceinline__ double wave_dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_sum(double v) {
#ifdef MCHAP_PLAIN_BUTTERFLY
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
#else
  v += __shfl_xor(v, 32, WAVE);
  v += __shfl_xor(v, 16, WAVE);
  v += wave_dpp_f64<0x128>(v);  // row_ror:8
  v += wave_dpp_f64<0x124>(v);  // row_ror:4
  v += wave_dpp_f64<0x122>(v);  // row_ror:2
  v += wave_dpp_f64<0x121>(v);  // row_ror:1
#endif
  return v;
}
// test hook (mchap_wave_sum_batch): one wavefront per 64 values; every lane must hold the same sum
static __global__ void wave_sum_kernel(const double *x, double *out) {
  const double s = wave_sum(x[(size_t)blockIdx.x * WAVE + threadIdx.x]);
  const bool same = __ballot(__double_as_longlong(s) == __double_as_longlong(__shfl(s, 0, WAVE))) == ~0ull;
  if (threadIdx.x == 0) out[blockIdx.x] = same ? s : __longlong_as_double(0x7ff8dead00000000ll);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

__device__ __forceinline__ uint32_t nib(uint32_t p, int h) { return (p >> (4 * h)) & 15u; }
__device__ __forceinline__ uint32_t nib_set(uint32_t p, int h, uint32_t v) {
  return (p & ~(15u << (4 * h))) | (v << (4 * h));
}
// ... and packs of sixteen nibbles (ploidies 9 to 15: the general lanes-over-chains sampler, denovo_simt_kernel<0, u128>)
__device__ __forceinline__ uint32_t nib(uint64_t p, int h) { return (uint32_t)(p >> (4 * h)) & 15u; }
__device__ __forceinline__ uint64_t nib_set(uint64_t p, int h, uint32_t v) {
  return (p & ~(15ull << (4 * h))) | ((uint64_t)v << (4 * h));
}

// jitutils.py:7-26
__device__ __forceinline__ double add_log_prob(double x, double y) {
  if (x == -INFINITY && y == -INFINITY) return -INFINITY;
  if (x > y) return x + log1p(exp(y - x));
  return y + log1p(exp(x - y));
}

// jitutils.py:77-92: searchsorted(cumsum(p), u, side="right")
__device__ __forceinline__ int choose_from(const double *p, int n, double u) {
  double c = 0.0;
  for (int i = 0; i < n; i++) {
    c += p[i];
    if (c > u) return i;
  }
  return n;
}

// first-occurrence dosage of the rows (in[h], out[h]) (jitutils.py:378-422 on label rows);
// a zero nibble marks a duplicate.
template <class P>
__device__ __forceinline__ P dosage_of_labels(P in, P out, int K, bool use_out) {
  P d = 0;
  for (int h = 0; h < K; h++) d |= (P)1 << (4 * h);
  for (int h = 0; h < K; h++) {
    if (nib(d, h) == 0) continue;
    for (int p = h + 1; p < K; p++) {
      if (nib(d, p) == 0) continue;
      if (nib(in, h) == nib(in, p) && (!use_out || nib(out, h) == nib(out, p))) {
        d += (P)1 << (4 * h);
        d &= ~((P)15 << (4 * p));
      }
    }
  }
  return d;
}

// structural.py:74-118
template <class P>
__device__ __forceinline__ int recombination_n_options(P in, P out, int K) {
  const P d = dosage_of_labels(in, out, K, true);
  int n = 0;
  for (int h0 = 0; h0 < K; h0++) {
    if (nib(d, h0) == 0) continue;
    for (int h1 = h0 + 1; h1 < K; h1++) {
      if (nib(d, h1) == 0) continue;
      if (nib(in, h0) == nib(in, h1) || nib(out, h0) == nib(out, h1)) continue;
      n++;
    }
  }
  return n;
}

// structural.py:181-237
template <class P>
__device__ __forceinline__ int dosage_n_options(P in, P out, int K) {
  const P hd = dosage_of_labels(in, out, K, true);
  const P sd = dosage_of_labels(in, out, K, false);
  int n = 0;
  for (int h0 = 0; h0 < K; h0++) {
    if (nib(hd, h0) == 0) continue;
    if (nib(sd, h0) == 1) continue;
    for (int h1 = 0; h1 < K; h1++) {
      if (nib(sd, h1) == 0) continue;
      if (nib(in, h0) == nib(in, h1)) continue;
      n++;
    }
  }
  return n;
}

// structural.py:121-178 / 240-307: option i = the `in` label pack after the move
__device__ inline int step_options(uint32_t in, uint32_t out, int K, int step_type, uint32_t *opt) {
  const uint32_t hd = dosage_of_labels(in, out, K, true);
  int n = 0;
  if (step_type == 0) {
    for (int h0 = 0; h0 < K; h0++) {
      if (nib(hd, h0) == 0) continue;
      for (int h1 = h0 + 1; h1 < K; h1++) {
        if (nib(hd, h1) == 0) continue;
        if (nib(in, h0) == nib(in, h1) || nib(out, h0) == nib(out, h1)) continue;
        uint32_t o = nib_set(in, h0, nib(in, h1));
        o = nib_set(o, h1, nib(in, h0));
        opt[n++] = o;
      }
    }
  } else {
    const uint32_t sd = dosage_of_labels(in, out, K, false);
    for (int h0 = 0; h0 < K; h0++) {
      if (nib(hd, h0) == 0) continue;
      if (nib(sd, h0) == 1) continue;
      for (int h1 = 0; h1 < K; h1++) {
        if (nib(sd, h1) == 0) continue;
        if (nib(in, h0) == nib(in, h1)) continue;
        opt[n++] = nib_set(in, h0, nib(in, h1));
      }
    }
  }
  return n;
}

// ---- per-chain context -------------------------------------------------------------------
struct Chain {
  // LDS
  const double *rl;   // staged reads [M0*A][rpad]
  uint64_t *w;        // [T][K] haplotype words per temperature
  uint64_t *pw;       // [K] proposal
  double *llk_t;      // [T]
  uint64_t *rngn;     // [T+1] draw counters
  double *probs, *llks;
  double *prior_tab;  // [0..K] lg(dose+disp) - (lg(dose+1)+lg(disp)); [K+1..2K+1] lg(dose+1); then left, lgK1, K*luh
  double *buf;
  uint32_t *optin;
  uint16_t *sub;
  uint16_t *hetrow;   // het position * A
  uint8_t *shift;
  uint8_t *nal;       // n_alleles of het positions
  // scalars
  int K, Mh, A, T, rpad, lane;
  uint32_t amask;
  int bits;
  double invK;
  double inbreeding;  // NaN == None
  double luh;
  Rng rng;
  uint64_t *cache;   // this chain's {tag, value} table or nullptr
  uint32_t cache_mask;
  int key_bits;      // bits per haplotype word actually used (Mh * bits)
  bool w01;          // every read weight of the unit is 0 or 1: one logarithm per group of four chunks (read_log_sum_chunks)
};

// Likelihood cache: the reference memoises log_likelihood per genotype in an array-backed trie
// (assemble/likelihood.py:151-305, arraymap.py) because a converged chain keeps proposing the same
// neighbours; it is results-neutral (identical output with the cache off).  Here: one direct-mapped
// table per chain in global memory (L2 / Infinity Cache resident), keyed by the exact packed genotype
// when it fits 63 bits, else by a 64-bit mix of the haplotype words; the stored value is the llk this
// same kernel computed for that key, so hits return bit-identical values.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
__device__ __forceinline__ uint64_t genotype_tag(const Chain &c, const uint64_t *hw) {
  uint64_t t = 0;
  if (c.key_bits * c.K <= 63) {
    for (int h = 0; h < c.K; h++) t = (t << c.key_bits) | hw[h];
  } else {
    for (int h = 0; h < c.K; h++) t = mix64(t ^ hw[h]) + 0x9E3779B97F4A7C15ull;
  }
  return (t << 1) | 1ull;
}

template <int RPL>
__device__ __forceinline__ double eval_llk(const Chain &c, const uint64_t *hw, const double (&cnt)[RPL]) {
  double acc[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) acc[i] = 0.0;
  for (int h = 0; h < c.K; h++) {
    const uint64_t wh = hw[h];
    double prod[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i++) prod[i] = 1.0;
    for (int j = 0; j < c.Mh; j++) {
      const uint32_t a = (uint32_t)(wh >> c.shift[j]) & c.amask;
      const double *row = c.rl + (size_t)(c.hetrow[j] + a) * c.rpad + c.lane;
#pragma unroll
      for (int i = 0; i < RPL; i++) prod[i] *= row[WAVE * i];
    }
#pragma unroll
    for (int i = 0; i < RPL; i++) acc[i] += prod[i] * c.invK;
  }
  return wave_sum(read_log_sum_chunks<RPL>(acc, cnt, c.w01));
}

template <int RPL>
__device__ __forceinline__ double eval_llk_cached(const Chain &c, const uint64_t *hw, const double (&cnt)[RPL]) {
  if (!c.cache || c.key_bits * c.K > 63) return eval_llk<RPL>(c, hw, cnt);  // (wide genotypes: not cached by this kernel)
  const uint64_t tag = genotype_tag(c, hw);
  const uint64_t slot = (c.key_bits * c.K <= 63 ? mix64(tag) : tag >> 1) & c.cache_mask;
  ulonglong2 *e = reinterpret_cast<ulonglong2 *>(c.cache) + slot;
  const ulonglong2 got = *e;  // one 16-byte load: {tag, llk bits}
  if (got.x == tag) return __longlong_as_double((long long)got.y);
  const double v = eval_llk<RPL>(c, hw, cnt);
  if (c.lane == 0) *e = make_ulonglong2(tag, (unsigned long long)__double_as_longlong(v));
  return v;
}

// dosage (first-occurrence convention, jitutils.py:378-422) of K haplotype words -> nibble pack
__device__ inline uint32_t dosage_of_words(const uint64_t *hw, int K) {
  uint32_t d = 0;
  for (int h = 0; h < K; h++) d |= 1u << (4 * h);
  for (int h = 0; h < K; h++) {
    if (nib(d, h) == 0) continue;
    for (int p = h + 1; p < K; p++) {
      if (nib(d, p) == 0) continue;
      if (hw[h] == hw[p]) {
        d += 1u << (4 * h);
        d &= ~(15u << (4 * p));
      }
    }
  }
  return d;
}

// jitutils.py:350-375
__device__ inline int count_copies(const uint64_t *hw, int K, int h) {
  int n = 0;
  const uint64_t x = hw[h];
  for (int i = 0; i < K; i++) n += (hw[i] == x) ? 1 : 0;
  return n;
}

// assemble/prior.py:81-112 from a dosage nibble pack (table built in the prologue)
__device__ inline double prior_of_dosage(const Chain &c, uint32_t d) {
  const double *t = c.prior_tab;
  const int K = c.K;
  if (c.inbreeding == 0.0) {
    // prior.py:15-36: lgamma(K+1) - sum lgamma(dose+1) - K*luh
    double den = 0.0;
    for (int i = 0; i < K; i++) den += t[K + 1 + nib(d, i)];
    return (t[2 * K + 3] - den) - t[2 * K + 4];
  }
  double prod = 0.0;
  for (int i = 0; i < K; i++) {
    const uint32_t dose = nib(d, i);
    if (dose > 0) prod += t[dose];
  }
  return t[2 * K + 2] + prod;
}

__device__ inline double words_prior(const Chain &c, const uint64_t *hw) {
  if (isnan(c.inbreeding)) return 0.0;
  return prior_of_dosage(c, dosage_of_words(hw, c.K));
}

// mutation.py:14-161
template <int RPL>
__device__ inline double base_step(Chain &c, uint64_t *wt, double llk, int h, int j, double temp,
                                   const double (&cnt)[RPL]) {
  const int K = c.K;
  const int n_alleles = c.nal[j];
  const int sh = c.shift[j];
  const double lhapcount = c_ln[count_copies(wt, K, h)];
  const double lprior = words_prior(c, wt);
  const uint64_t wh = wt[h];
  const int current = (int)((wh >> sh) & c.amask);
  int n_options = 0;
  double *la = c.probs;  // log_accept, then probabilities
  for (int i = 0; i < K; i++) c.pw[i] = wt[i];
  for (int i = 0; i < n_alleles; i++) {
    if (i == current) {
      c.llks[i] = llk;
      la[i] = -INFINITY;
    } else {
      n_options += 1;
      c.pw[h] = (wh & ~((uint64_t)c.amask << sh)) | ((uint64_t)i << sh);
      const double llk_i = eval_llk_cached<RPL>(c, c.pw, cnt);
      c.llks[i] = llk_i;
      const double llk_ratio = llk_i - llk;
      double lprior_ratio = 0.0;
      if (!isnan(c.inbreeding)) lprior_ratio = words_prior(c, c.pw) - lprior;
      const double lproposal_ratio = c_ln[count_copies(c.pw, K, h)] - lhapcount;
      const double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
      la[i] = fmin(0.0, mh);
    }
  }
  const double ln_opt = c_ln[n_options];
  double sum = 0.0;
  for (int i = 0; i < n_alleles; i++) {
    const double p = exp(la[i] - ln_opt);
    la[i] = p;
    sum += p;
  }
  la[current] = 1.0 - sum;
  int choice = choose_from(la, n_alleles, rng_double(c.rng));
  if (choice >= n_alleles) choice = n_alleles - 1;
  wt[h] = (wh & ~((uint64_t)c.amask << sh)) | ((uint64_t)choice << sh);
  return c.llks[choice];
}

// mutation.py:164-246
template <int RPL>
__device__ inline double mutation_compound_step(Chain &c, uint64_t *wt, double llk, double temp,
                                                const double (&cnt)[RPL]) {
  const int n = c.K * c.Mh;
  for (int i = 0; i < n; i++) c.sub[i] = (uint16_t)i;
  for (int i = n - 1; i >= 1; i--) {
    const int k = (int)rng_interval(c.rng, (uint32_t)i);
    const uint16_t a = c.sub[i], b = c.sub[k];
    c.sub[i] = b;
    c.sub[k] = a;
  }
  for (int i = 0; i < n; i++) {
    const int s = c.sub[i];
    llk = base_step<RPL>(c, wt, llk, s / c.Mh, s % c.Mh, temp, cnt);
  }
  return llk;
}

__device__ inline uint64_t interval_mask(const Chain &c, int start, int stop) {
  // fields of positions start..stop-1; position j sits at bit shift[j] = bits*(Mh-1-j)
  const int nb = c.bits * (stop - start);
  const uint64_t ones = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
  return ones << (c.bits * (c.Mh - stop));
}

// first-occurrence labels of the haplotype segments selected by `mask` (structural.py:310-430)
__device__ inline uint32_t segment_labels(const uint64_t *hw, int K, uint64_t mask) {
  uint32_t lab = 0;
  for (int h = 1; h < K; h++) {
    int l = h;
    for (int g = 0; g < h; g++) {
      if (((hw[g] ^ hw[h]) & mask) == 0) {
        l = g;
        break;
      }
    }
    lab |= (uint32_t)l << (4 * h);
  }
  return lab;
}

// structural.py:433-587
template <int RPL>
__device__ inline double interval_step(Chain &c, uint64_t *wt, double llk, int start, int stop, int step_type,
                                       double temp, const double (&cnt)[RPL]) {
  const int K = c.K;
  const uint64_t full = interval_mask(c, 0, c.Mh);
  const uint64_t min_ = interval_mask(c, start, stop);
  const uint32_t lin = segment_labels(wt, K, min_);
  const uint32_t lout = segment_labels(wt, K, full & ~min_);
  const int n_options = step_options(lin, lout, K, step_type, c.optin);
  if (n_options == 0) return llk;
  const double log_proposal_prob = c_ln_inv[n_options];
  double lprior = 0.0;
  if (!isnan(c.inbreeding)) lprior = words_prior(c, wt);
  double *la = c.probs;
  c.llks[n_options] = -INFINITY;
  la[n_options] = -INFINITY;
  for (int i = 0; i < n_options; i++) {
    const uint32_t oin = c.optin[i];
    for (int h = 0; h < K; h++) c.pw[h] = (wt[h] & ~min_) | (wt[nib(oin, h)] & min_);
    const double llk_i = eval_llk_cached<RPL>(c, c.pw, cnt);
    c.llks[i] = llk_i;
    const double llk_ratio = llk_i - llk;
    double lprior_ratio = 0.0;
    if (!isnan(c.inbreeding)) lprior_ratio = prior_of_dosage(c, dosage_of_labels(oin, lout, K, true)) - lprior;
    const int n_return = step_type == 0 ? recombination_n_options(oin, lout, K) : dosage_n_options(oin, lout, K);
    const double lproposal_ratio = c_ln_inv[n_return] - log_proposal_prob;
    const double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
    la[i] = fmin(0.0, mh);
  }
  const double ln_opt = c_ln[n_options];
  double sum = 0.0;
  for (int i = 0; i <= n_options; i++) {
    const double p = exp(la[i] - ln_opt);
    la[i] = p;
    sum += p;
  }
  la[n_options] = 1.0 - sum;
  const int choice = choose_from(la, n_options + 1, rng_double(c.rng));
  if (choice < n_options) {
    const uint32_t oin = c.optin[choice];
    for (int h = 0; h < K; h++) c.pw[h] = (wt[h] & ~min_) | (wt[nib(oin, h)] & min_);
    for (int h = 0; h < K; h++) wt[h] = c.pw[h];
    llk = c.llks[choice];
  }
  return llk;
}

// structural.py:22-71 + 590-673.  Returns <0 on the reference's ValueError.
template <int RPL>
__device__ inline int structural_compound_step(Chain &c, uint64_t *wt, double &llk, int n_breaks, bool whole,
                                               int step_type, double temp, const double (&cnt)[RPL]) {
  const int n = c.Mh;
  int *points = reinterpret_cast<int *>(c.buf);
  int n_int;
  if (whole) {
    points[0] = 0;
    points[1] = n;
    n_int = 1;
  } else {
    if (n_breaks >= n) return -1;
    // indicator bits 1..n-1 are candidate break points
    uint64_t ind = 0;
    for (int i = 1; i < n; i++) ind |= 1ull << i;
    for (int b = 0; b < n_breaks; b++) {
      const int no = __popcll(ind);
      if (no == 0) break;
      int k = (int)rng_interval(c.rng, (uint32_t)(no - 1));
      uint64_t t = ind;
      while (k-- > 0) t &= t - 1;  // drop the k lowest candidates
      ind &= ~(t & (~t + 1));       // clear the k-th
    }
    uint64_t zeros = ~ind & ((n >= 63 ? ~0ull : ((1ull << (n + 1)) - 1ull)));
    int np = 0;
    while (zeros) {
      points[np++] = __ffsll((long long)zeros) - 1;
      zeros &= zeros - 1;
    }
    n_int = n_breaks + 1;
  }
  // np.random.permutation(arange(n_int)): Fisher-Yates on a small order array kept in c.sub
  for (int i = 0; i < n_int; i++) c.sub[i] = (uint16_t)i;
  for (int i = n_int - 1; i >= 1; i--) {
    const int k = (int)rng_interval(c.rng, (uint32_t)i);
    const uint16_t a = c.sub[i], b = c.sub[k];
    c.sub[i] = b;
    c.sub[k] = a;
  }
  for (int i = 0; i < n_int; i++) {
    const int iv = c.sub[i];
    llk = interval_step<RPL>(c, wt, llk, points[iv], points[iv + 1], step_type, temp, cnt);
  }
  return 0;
}

// tempering.py:10-151; i = cooler (current), j = warmer (previous)
__device__ inline void chain_swap_step(Chain &c, uint64_t *wi, double &llk_i, double temp_i, uint64_t *wj,
                                       double &llk_j, double temp_j) {
  const double prior_i = words_prior(c, wi);
  const double prior_j = words_prior(c, wj);
  const double ui = llk_i + prior_i, uj = llk_j + prior_j;
  double acc = exp((uj - ui) * temp_i + (ui - uj) * temp_j);
  if (acc > 1.0) acc = 1.0;
  const double val = rng_double(c.rng);
  if (acc >= val) {
    for (int h = 0; h < c.K; h++) {
      const uint64_t t = wi[h];
      wi[h] = wj[h];
      wj[h] = t;
    }
    const double t = llk_i;
    llk_i = llk_j;
    llk_j = t;
  }
}

// calling/prior.py:116-179 with frequencies=None, on a nibble-packed SNV genotype (snpcalling.py:55-60)
__device__ inline double snv_log_prior(uint32_t g, int K, int n_alleles, double F) {
  // allelic dosage, first-occurrence convention (calling/utils.py:7-35)
  int dose[MCHAP_MAX_PLOIDY];
  for (int i = 0; i < K; i++) dose[i] = 0;
  for (int i = 0; i < K; i++) {
    int j = 0;
    while (nib(g, i) != nib(g, j)) j++;
    dose[j] += 1;
  }
  if (F == 0.0) {
    double den = 0.0;
    for (int i = 0; i < K; i++) den += lgamma((double)dose[i] + 1.0);
    return (lgamma((double)K + 1.0) - den) - (double)K * c_ln[n_alleles];
  }
  const double alpha = (1.0 / (double)n_alleles) * ((1.0 - F) / F);
  const double sum_alphas = alpha * (double)n_alleles;
  const double left = (lgamma((double)K + 1.0) + lgamma(sum_alphas)) - lgamma((double)K + sum_alphas);
  double prod = 0.0;
  for (int i = 0; i < K; i++) {
    if (dose[i] > 0) prod += lgamma((double)dose[i] + alpha) - (lgamma((double)dose[i] + 1.0) + lgamma(alpha));
  }
  return left + prod;
}

// jitutils.py:114-146 on a nibble pack
template <class P>
__device__ inline P increment_snv_genotype(P g, int K) {
  if (K == 1) return g + 1;
  const uint32_t previous = nib(g, 0);
  for (int i = 1; i < K; i++) {
    const uint32_t allele = nib(g, i);
    if (allele == previous) continue;
    // allele > previous
    const int k = i - 1;
    g = nib_set(g, k, nib(g, k) + 1);
    for (int z = 0; z < k; z++) g = nib_set(g, z, 0);
    return g;
  }
  g = nib_set(g, K - 1, nib(g, K - 1) + 1);
  for (int z = 0; z < K - 1; z++) g = nib_set(g, z, 0);
  return g;
}

constexpr int CHAINS_PER_BLOCK = 4;

// grid = (units, ceil(chains / CHAINS_PER_BLOCK)); block = 64 * min(chains, CHAINS_PER_BLOCK)
template <int RPL>
__global__ __launch_bounds__(64 * CHAINS_PER_BLOCK) void denovo_mcmc_kernel(const DenovoParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const mchap_unit U = P.units[blockIdx.x];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chain = blockIdx.y * CHAINS_PER_BLOCK + wave;
  const int R = U.n_reads, M0 = U.n_pos, A = U.max_allele, K = U.ploidy, T = P.n_temps;
  const int rpad = P.rpad;
  const int S = P.steps;

  double *rl = reinterpret_cast<double *>(smem);
  const size_t rl_bytes = (size_t)M0 * A * rpad * sizeof(double);
  const WaveLayout L = wave_layout(K, M0, A, T);
  unsigned char *ws = smem + rl_bytes + (size_t)wave * L.total;

  // ---- stage reads: [R][M0][A] (HBM) -> [M0*A][rpad] (LDS), NaN -> 1.0 ----
  const double *gr = P.reads + U.reads_off;
  const int MA = M0 * A;
  for (int r = threadIdx.x; r < rpad; r += blockDim.x) {
    if (r < R) {
      const double *src = gr + (size_t)r * MA;
      for (int q = 0; q < MA; q++) {
        const double v = src[q];
        rl[(size_t)q * rpad + r] = isnan(v) ? 1.0 : v;
      }
    } else {
      for (int q = 0; q < MA; q++) rl[(size_t)q * rpad + r] = 1.0;
    }
  }
  double cnt[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) {
    const int r = lane + WAVE * i;
    cnt[i] = (r < R) ? (U.counts_off >= 0 ? (double)P.counts[U.counts_off + r] : 1.0) : 0.0;
  }
  bool w01_l = true;
#pragma unroll
  for (int i = 0; i < RPL; i++) w01_l = w01_l && (cnt[i] == 0.0 || cnt[i] == 1.0);
  const bool w01_u = __ballot(!w01_l) == 0ull;  // (as the prepare pass of the other kernels decides: META_I_W01)
  __syncthreads();
  if (chain >= P.chains) return;

  Chain c;
  c.w01 = w01_u;
  c.rl = rl;
  c.w = reinterpret_cast<uint64_t *>(ws + L.w);
  c.pw = reinterpret_cast<uint64_t *>(ws + L.pw);
  c.llk_t = reinterpret_cast<double *>(ws + L.llk);
  c.rngn = reinterpret_cast<uint64_t *>(ws + L.rngn);
  c.probs = reinterpret_cast<double *>(ws + L.probs);
  c.llks = reinterpret_cast<double *>(ws + L.llks);
  c.prior_tab = reinterpret_cast<double *>(ws + L.prior);
  c.buf = reinterpret_cast<double *>(ws + L.buf);
  c.optin = reinterpret_cast<uint32_t *>(ws + L.optin);
  c.sub = reinterpret_cast<uint16_t *>(ws + L.sub);
  c.hetrow = reinterpret_cast<uint16_t *>(ws + L.hetrow);
  c.shift = ws + L.shift;
  c.nal = ws + L.nal;
  c.K = K;
  c.A = A;
  c.T = T;
  c.rpad = rpad;
  c.lane = lane;
  c.invK = 1.0 / (double)K;
  c.inbreeding = U.inbreeding;
  c.bits = allele_bits(A);
  c.amask = (1u << c.bits) - 1u;

  const int8_t *nalleles = P.n_alleles + U.nalleles_off;

  // ---- homozygous fix (assemble/mcmc.py:168-182, 494-541; snpcalling.py:14-70) ----
  int Mh = 0;
  for (int j = 0; j < M0; j++) {
    const int n = nalleles[j];
    const int u_gens = snv_genotypes(n, K);
    uint32_t g = 0;
    double *lp = c.buf;
    for (int q = 0; q < u_gens; q++) {
      double lprior = 0.0;
      if (!isnan(U.inbreeding)) lprior = snv_log_prior(g, K, n, U.inbreeding);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < RPL; i++) {
        double rp = 0.0;
        for (int h = 0; h < K; h++) rp += rl[(size_t)(j * A + nib(g, h)) * rpad + lane + WAVE * i] / (double)K;
        s += read_log(rp) * cnt[i];
      }
      const double llk = wave_sum(s);
      lp[q] = lprior + llk;
      g = increment_snv_genotype(g, K);
    }
    // normalise_log_probs (jitutils.py:51-74) and the homozygous genotypes' probabilities
    double acc = lp[0];
    for (int q = 1; q < u_gens; q++) acc = add_log_prob(acc, lp[q]);
    int fixed_allele = -1;
    for (int a = 0; a < n; a++) {
      // index of genotype (a, a, ..., a): sum_i C(a + i, i + 1)  (jitutils.py:253-276)
      int idx = 0;
      for (int i = 0; i < K; i++) idx += (a == 0) ? 0 : snv_genotypes(a, i + 1);
      const double p = exp(lp[idx] - acc);
      if (p >= P.fix_hom) fixed_allele = a;
    }
    if (chain == 0 && lane == 0) P.fixed[U.fixed_off + j] = (int8_t)fixed_allele;
    if (fixed_allele < 0) {
      c.hetrow[Mh] = (uint16_t)(j * A);
      c.nal[Mh] = (uint8_t)n;
      Mh++;
    }
  }
  c.Mh = Mh;
  const size_t trace_base = U.trace_off + (size_t)chain * S * K;
  const size_t llk_base = U.llk_off + (size_t)chain * S;
  if (Mh == 0) {
    // assemble/mcmc.py:189-199
    if (chain == 0 && lane == 0) P.status[blockIdx.x] = MCHAP_UNIT_ALL_FIXED;
    for (int i = lane; i < S * K; i += WAVE) P.trace[trace_base + i] = 0ull;
    for (int i = lane; i < S; i += WAVE) P.llks[llk_base + i] = NAN;
    return;
  }
  if (Mh * c.bits > 64) {
    if (chain == 0 && lane == 0) P.status[blockIdx.x] = MCHAP_ERR_LIMIT;
    return;
  }
  if (U.initial_off >= 0 && U.initial_n_het != Mh) {  // assemble/mcmc.py:207
    if (chain == 0 && lane == 0) P.status[blockIdx.x] = MCHAP_UNIT_BAD_INITIAL;
    return;
  }
  double luh = 0.0;
  for (int j = 0; j < Mh; j++) {
    c.shift[j] = (uint8_t)(c.bits * (Mh - 1 - j));
    luh += c_ln[c.nal[j]];  // assemble/mcmc.py:294
  }
  c.luh = luh;
  c.key_bits = c.bits * Mh;
  c.cache = nullptr;
  c.cache_mask = 0;
  if (P.cache_slots > 0) {
    c.cache = P.cache + ((size_t)blockIdx.x * P.chains + chain) * (size_t)P.cache_slots * 2;
    c.cache_mask = (uint32_t)P.cache_slots - 1u;
  }
  if (!isnan(U.inbreeding)) {
    // tables for assemble/prior.py:39-112
    double *t = c.prior_tab;
    const double F = U.inbreeding;
    for (int d = 0; d <= K; d++) t[K + 1 + d] = lgamma((double)d + 1.0);
    t[2 * K + 3] = lgamma((double)K + 1.0);
    t[2 * K + 4] = (double)K * luh;
    if (F != 0.0) {
      const double log_disp = log((1.0 - F) / F) - luh;
      const double disp = exp(log_disp);
      const double sum_disp = exp(log_disp + luh);
      const double lg_disp = lgamma(disp);
      t[0] = 0.0;
      for (int d = 1; d <= K; d++) t[d] = lgamma((double)d + disp) - (lgamma((double)d + 1.0) + lg_disp);
      t[2 * K + 2] = (lgamma((double)K + 1.0) + lgamma(sum_disp)) - lgamma((double)K + sum_disp);
    }
  }

  // ---- initial genotype (assemble/mcmc.py:202-208) ----
  uint64_t *w0 = c.w;  // temperature 0 slot used as staging; copied to every temperature below
  if (U.initial_off >= 0) {
    const int8_t *ini = P.initial + U.initial_off + (size_t)chain * K * Mh;
    for (int h = 0; h < K; h++) {
      uint64_t x = 0;
      for (int j = 0; j < Mh; j++) x |= (uint64_t)(uint8_t)ini[h * Mh + j] << c.shift[j];
      w0[h] = x;
    }
  } else {
    // _read_mean_dist (assemble/mcmc.py:455-491) from the raw tensor (NaN information needed)
    double *dist = c.buf;
    for (int j = 0; j < Mh; j++) {
      const int col = c.hetrow[j];
      int n_nonzero = 0;
      uint32_t gapmask = 0;
      for (int a = 0; a < A; a++) {
        double tot = 0.0;
        int n_ok = 0, n_nz = 0;
        for (int r = lane; r < R; r += WAVE) {
          const double v = gr[(size_t)r * MA + col + a];
          if (!isnan(v)) {
            tot += v;
            n_ok++;
          }
          if (!(v == 0.0)) n_nz++;  // NaN != 0
        }
        tot = wave_sum(tot);
        n_ok = wave_sum_i(n_ok);
        n_nz = wave_sum_i(n_nz);
        if (n_ok == 0) {
          // all-gap column: replaced by ones (mcmc.py:480) -> mean 1, not identically zero
          gapmask |= 1u << a;
          dist[j * A + a] = 1.0;
          n_nonzero++;
        } else {
          dist[j * A + a] = tot / (double)n_ok;
          if (n_nz > 0) n_nonzero++;
        }
      }
      double s = 0.0;
      for (int a = 0; a < A; a++) {
        if (gapmask & (1u << a)) dist[j * A + a] = 1.0 / (double)n_nonzero;
      }
      // numpy add.reduce over the last axis: first element + sequential rest
      for (int a = 1; a < A; a++) s += dist[j * A + a];
      s = (A > 1) ? dist[j * A] + s : dist[j * A];
      for (int a = 0; a < A; a++) dist[j * A + a] /= s;
    }
    rng_open(c.rng, P.seed, U.stream_id, (uint32_t)chain, SLOT_INIT, 0);
    for (int h = 0; h < K; h++) {
      uint64_t x = 0;
      for (int j = 0; j < Mh; j++) {
        // sample_snv_alleles (jitutils.py:464-498)
        double s = 0.0;
        for (int a = 0; a < A; a++) s += dist[j * A + a];
        double cacc = 0.0;
        const double u = rng_double(c.rng);
        int ch = A;
        for (int a = 0; a < A; a++) {
          cacc += dist[j * A + a] / s;
          if (cacc > u) {
            ch = a;
            break;
          }
        }
        if (ch >= A) ch = A - 1;
        x |= (uint64_t)ch << c.shift[j];
      }
      w0[h] = x;
    }
  }
  {
    const double llk0 = eval_llk<RPL>(c, w0, cnt);  // assemble/mcmc.py:303
    for (int t = T - 1; t >= 0; t--) {
      for (int h = 0; h < K; h++) c.w[t * K + h] = w0[h];
      c.llk_t[t] = llk0;
      c.rngn[t] = 0;
    }
  }
  const double *break_dist = P.break_table + (size_t)Mh * P.max_pos;
  const int n_break_dist = P.n_intervals > 0 ? P.n_intervals : Mh;

  // ---- main loop (assemble/mcmc.py:323-425) ----
  int status = MCHAP_UNIT_OK;
  for (int step = 0; step < S && status == MCHAP_UNIT_OK; step++) {
    for (int t = 0; t < T; t++) {
      uint64_t *wt = c.w + t * K;
      double llk = c.llk_t[t];
      const double temp = P.temps[t];
      if (isnan(llk)) {
        status = MCHAP_UNIT_NAN_LLK;
        break;
      }
      rng_open(c.rng, P.seed, U.stream_id, (uint32_t)chain, (uint32_t)t, (uint64_t)step * STEP_DRAWS);
      llk = mutation_compound_step<RPL>(c, wt, llk, temp, cnt);
      for (int kind = 0; kind < 2; kind++) {
        const double pstep = kind == 0 ? P.p_recomb : P.p_partial;
        if (rng_double(c.rng) <= pstep) {
          int nb;
          if (P.n_intervals > 0) {
            // break_dist = [0,...,0,1] (assemble/mcmc.py:214-217): the draw is still consumed
            (void)rng_double(c.rng);
            nb = P.n_intervals - 1;
          } else {
            nb = choose_from(break_dist, n_break_dist, rng_double(c.rng));
          }
          if (structural_compound_step<RPL>(c, wt, llk, nb, false, kind, temp, cnt) < 0) {
            status = MCHAP_UNIT_BREAKS;
            break;
          }
        }
      }
      if (status != MCHAP_UNIT_OK) break;
      if (rng_double(c.rng) <= P.p_dosage) structural_compound_step<RPL>(c, wt, llk, 0, true, 1, temp, cnt);
      if (t > 0) {
        double llk_prev = c.llk_t[t - 1];
        chain_swap_step(c, wt, llk, temp, c.w + (t - 1) * K, llk_prev, P.temps[t - 1]);
        c.llk_t[t - 1] = llk_prev;
      }
      c.llk_t[t] = llk;
      c.rngn[t] = c.rng.n;
    }
    if (status != MCHAP_UNIT_OK) break;
    // record the cold chain, haplotypes in canonical (ascending) order (assemble/classes.py:265-278)
    const uint64_t *wc = c.w + (T - 1) * K;
    if (lane < K) {
      const uint64_t x = wc[lane];
      int rank = 0;
      for (int h = 0; h < K; h++) {
        const uint64_t y = wc[h];
        rank += (y < x || (y == x && h < lane)) ? 1 : 0;
      }
      P.trace[trace_base + (size_t)step * K + rank] = x;
    }
    if (lane == 0) P.llks[llk_base + step] = c.llk_t[T - 1];
  }
  if (lane == 0 && status != MCHAP_UNIT_OK) atomicMax(&P.status[blockIdx.x], status);
}

// Test hook: log_likelihood (assemble/likelihood.py:17-70) of many genotypes of one unit; one wavefront
// per genotype, same staging and evaluation code as the sampler.
template <int RPL>
__global__ __launch_bounds__(256) void llk_batch_kernel(const double *reads, int R, int M, int A, const int64_t *counts,
                                                        const int8_t *genotypes, int n_genotypes, int K, int rpad,
                                                        double *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *rl = reinterpret_cast<double *>(smem);
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int MA = M * A;
  for (int r = threadIdx.x; r < rpad; r += blockDim.x) {
    for (int q = 0; q < MA; q++) {
      const double v = r < R ? reads[(size_t)r * MA + q] : 1.0;
      rl[(size_t)q * rpad + r] = isnan(v) ? 1.0 : v;
    }
  }
  double cnt[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) {
    const int r = lane + WAVE * i;
    cnt[i] = (r < R) ? (counts ? (double)counts[r] : 1.0) : 0.0;
  }
  bool w01_l = true;
#pragma unroll
  for (int i = 0; i < RPL; i++) w01_l = w01_l && (cnt[i] == 0.0 || cnt[i] == 1.0);
  unsigned char *ws = smem + (size_t)MA * rpad * sizeof(double) + (size_t)wave * (8 * K + 4 * M + 64);
  Chain c;
  c.w01 = __ballot(!w01_l) == 0ull;  // (the samplers' rule: one logarithm per group of four chunks where the weights are 0 / 1)
  c.rl = rl;
  c.pw = reinterpret_cast<uint64_t *>(ws);
  c.hetrow = reinterpret_cast<uint16_t *>(ws + 8 * K);
  c.shift = ws + 8 * K + align8(2 * M);
  c.K = K;
  c.Mh = M;
  c.A = A;
  c.rpad = rpad;
  c.lane = lane;
  c.invK = 1.0 / (double)K;
  c.bits = allele_bits(A);
  c.amask = (1u << c.bits) - 1u;
  for (int j = 0; j < M; j++) {
    c.hetrow[j] = (uint16_t)(j * A);
    c.shift[j] = (uint8_t)(c.bits * (M - 1 - j));
  }
  __syncthreads();
  const int nw = blockDim.x / WAVE;
  for (int g = blockIdx.x * nw + wave; g < n_genotypes; g += gridDim.x * nw) {
    const int8_t *gt = genotypes + (size_t)g * K * M;
    for (int h = 0; h < K; h++) {
      uint64_t x = 0;
      for (int j = 0; j < M; j++) x |= (uint64_t)(uint8_t)gt[h * M + j] << c.shift[j];
      c.pw[h] = x;
    }
    const double llk = eval_llk<RPL>(c, c.pw, cnt);
    if (lane == 0) out[g] = llk;
  }
}

}  // namespace mchap
