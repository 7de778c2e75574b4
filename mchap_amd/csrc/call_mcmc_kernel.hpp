// `mchap call`: Gibbs / Metropolis-Hastings sampler over the genotypes of KNOWN haplotypes, for MI355X (gfx950).
// Reference: calling/mcmc.py:15-453 (mh_options, gibbs_options, compound_step, mcmc_sampler, greedy_caller),
// calling/classes.py:14-124 (CallingMCMC.fit), calling/prior.py:30-179, calling/likelihood.py:8-78, calling/utils.py:36-58.
//
// One wavefront per (unit, chain).  The per-(read, haplotype) products P[r][h] of the exact caller (exact_kernel.hpp) sit in
// LDS, so a likelihood is R * K table reads; a sub-step's options (one per known haplotype) are spread over the lanes:
// each lane forms its proposal's key (VCF index of the sorted alleles) and probes the chain's table of remembered
// likelihoods -- the reference's dict (calling/likelihood.py:36-78: keyed on the SORTED genotype, so the first
// evaluated allele order's value is what every later request gets; the table here never evicts, to keep that) -- and the
// misses are evaluated one after the other by the whole wavefront, lanes over reads.  The categorical draw follows the
// reference's sequential arithmetic (normalise_log_probs / cumsum in allele order) on one lane.
// Philox streams as in philox.hpp: key (seed, unit stream), counter word 2 = chain << 16, draws in the reference's order: K - 1 shuffle draws, then one uniform per sub-step.
#pragma once
#include "exact_kernel.hpp"

namespace mchap {

constexpr uint32_t SLOT_CALL = 0u;  // the chain's stream (the de novo sampler's temperature-0 slot: another operator, another call)

// stateless Philox draws of a stream (the contract of philox.hpp)
struct CallStream {
  uint32_t k0, k1, c2, c3;
};
__device__ __forceinline__ void call_words(const CallStream &s, uint64_t n, uint32_t &a, uint32_t &b) {
  uint32_t o[4];
  const uint64_t blk = n >> 1;
  philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s.c2, s.c3, s.k0, s.k1, o);
  a = (n & 1) ? o[2] : o[0];
  b = (n & 1) ? o[3] : o[1];
}
__device__ __forceinline__ double call_double(const CallStream &s, uint64_t n) {
  uint32_t a, b;
  call_words(s, n, a, b);
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ uint32_t call_interval(const CallStream &s, uint64_t n, uint32_t max) {
  uint32_t a, b;
  call_words(s, n, a, b);
  return __umulhi(a, max + 1u);
}
constexpr int CALL_MAX_HAPS = 256;

struct CallParams {
  const double *reads;       // [U][R][M][A]
  const int64_t *counts;     // [U][R] or null
  const int8_t *haps;        // [U][H][M]
  const double *inbreeding;  // [U] or null (no prior)
  const double *freqs;       // [U][H] or null
  const int64_t *initial;    // [U][K] or null (greedy_caller)
  const uint64_t *stream_ids;  // [U]
  int R, M, A, H, K;
  int has_prior, step_type, steps, chains;
  uint64_t seed;
  // per-chain table of remembered likelihoods: [U * chains][cache_slots] of {key + 1, llk bits}; never evicts
  ulonglong2 *cache;
  long long cache_slots;     // power of two
  double *ptab_ext;          // [U * chains][R * H + R] product tables in the workspace when R * H * 8 exceeds the LDS, else null
  int64_t *genotypes;        // [U][chains][steps][K] sorted alleles
  double *llks;              // [U][chains][steps]
  int32_t *status;           // [U]: 0 ok, MCHAP_ERR_LIMIT if a table filled up
  // settled chains on their own kernel (call_coast_kernel below): the chain's hand-over record in the workspace, or null
  uint64_t *state;           // [U * chains][state_stride] words (call_state_words)
  int state_stride;
  int phase;                 // 0: a chain's start; 1: the chains call_coast_kernel handed back (CALL_RESUME)
  int last;                  // run every chain to its end: no hand-over
  int n_units;
};

// A chain's hand-over record: word 0 = next step | flag << 32, word 1 = entry the next memo miss replaces, words 2-5 the genotype
// at that step's start (eight alleles of 32 bits), then the Gibbs memo as it sits in LDS: CALL_MEMO keys, the entries' H
// probabilities and H likelihoods.
constexpr int CALL_STATE_HDR = 6;
constexpr unsigned CALL_START = 0u, CALL_COAST = 1u, CALL_RESUME = 2u, CALL_DONE = 3u;

// Memo of Gibbs sub-steps (round 3): the option probabilities of a sub-step are a function of the OTHER K - 1 alleles of the
// genotype alone -- the likelihood table never forgets or changes an entry, the priors are tables -- and normalising them is
// a chain of H - 1 dependent add_log_prob (an exp and a log1p each: calling/mcmc.py via jitutils.py:30-74) that only one
// lane can run: measured 36 us of a 37 us sub-step.  A converged chain meets the same few contexts step after step, so the
// last CALL_MEMO contexts keep their probabilities and likelihoods in LDS ([entries][2 H] doubles + keys); a hit goes straight
// to the draw.  Same values by construction.
constexpr int CALL_MEMO = 8;
__host__ __device__ inline int call_memo_entries(int H) {
  const int per = 2 * H * 8;  // bytes of an entry's probabilities and likelihoods
  const int n = 16384 / per;
  return n > CALL_MEMO ? CALL_MEMO : n;
}
__host__ __device__ inline int call_state_words(int H) { return CALL_STATE_HDR + CALL_MEMO + call_memo_entries(H) * 2 * H; }
// A workgroup is up to CALL_WG_CHAINS wavefronts: the chains of ONE unit, which share the unit's tables (product table, read
// weights, prior tables: 28 of the 32 KB a chain needed at the `mchap call` bench shape -- the LDS, not the registers, set the
// occupancy: 4 wavefronts per CU with a table per chain, 8 with two chains per table).  Each chain keeps its own option arrays,
// request words and Gibbs memo behind the shared part.
#ifndef MCHAP_CALL_WG_CHAINS
#define MCHAP_CALL_WG_CHAINS 4
#endif
constexpr int CALL_WG_CHAINS = MCHAP_CALL_WG_CHAINS;
// doubles of a chain's private part: 4 arrays [H] + request words + the Gibbs memo
__host__ __device__ inline size_t call_lds_private(int H) { return 4 * (size_t)H + 64 + (size_t)CALL_MEMO + (size_t)call_memo_entries(H) * 2 * H; }
inline size_t call_lds_bytes(int R, int H, int K, int wg_chains = 1) {
  // shared: ptab, cnt, lgd, lgf, lfreq (exact_setup) + rtab [H][K]
  return ((size_t)R * H + R + (size_t)H * (K + 1) + (K + 1) + H + (size_t)H * K + (size_t)wg_chains * call_lds_private(H)) * 8;
}
// LDS hand-over between the lanes of ONE wavefront (the chains of a workgroup run apart).  The fence names the LDS only: a
// fence over all address spaces also waits for the step's trace stores to HBM (measured: 150 -> 194 ms per call).
__device__ __forceinline__ void call_sync() {
#ifdef MCHAP_CALL_SYNC_BARRIER
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
#endif
}
// ... and of the chain's table of remembered likelihoods in HBM (written by one lane, probed by all)
__device__ __forceinline__ void call_sync_global() {
#ifdef MCHAP_CALL_SYNC_BARRIER
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#endif
}

// calling/prior.py:116-179 on the alleles in ARRAY order (allelic dosage at first occurrence, calling/utils.py:7-35)
__device__ inline double calling_log_prior_unsorted(const PriorTab &t, const int *g, int K) {
  if (t.F == 0.0) {
    double den = 0.0;
    for (int i = 0; i < K; i++) {
      bool first = true;
      int dose = 0;
      for (int j = 0; j < K; j++) {
        if (g[j] == g[i]) {
          dose++;
          if (j < i) first = false;
        }
      }
      den += first ? t.lgf[dose] : 0.0;  // lgamma(0 + 1) = 0 at the later copies
    }
    const double ln_perms = t.lgf[K] - den;
    if (!t.has_freqs) return ln_perms - (double)K * t.lnH;
    double prod = 1.0;
    for (int q = 0; q < K; q++) prod *= t.lfreq[g[q]];
    return ln_perms + log(prod);
  }
  double prod = 0.0;
  for (int i = 0; i < K; i++) {
    bool first = true;
    int dose = 0;
    for (int j = 0; j < K; j++) {
      if (g[j] == g[i]) {
        dose++;
        if (j < i) first = false;
      }
    }
    if (first) prod += t.lgd[g[i] * (K + 1) + dose];
  }
  return t.left + prod;
}

__device__ __forceinline__ long long call_key(const int *g, int K) {
  int s[EXACT_KMAX];
  for (int i = 0; i < K; i++) s[i] = g[i];
  for (int a = 1; a < K; a++) {  // insertion sort
    const int v = s[a];
    int b = a - 1;
    while (b >= 0 && s[b] > v) {
      s[b + 1] = s[b];
      b--;
    }
    s[b + 1] = v;
  }
  return rank_genotype(s, K);
}

// Key of a Gibbs context over at most sixteen known haplotypes: how many copies of each the other K - 1 <= 7 alleles hold, four
// bits a haplotype -- one-to-one on the multiset, never 0 for K >= 2 (0 marks an empty memo entry), and a handful of shifts
// where sorting the alleles costs a network
template <int KM, typename Get>
__device__ __forceinline__ unsigned long long call_ctx_counts(Get allele, int K, int k) {
  unsigned long long key = 0ull;
#pragma unroll
  for (int i = 0; i < KM; i++)
    if (i < K && i != k) key += 1ull << (4 * (allele(i) & 15));
  return key;
}

// KM: the ploidy bound of the genotype arrays and unrolled loops (8, or EXACT_KMAX = 16 for ploidies 9 to 15: exact_kernel.hpp)
template <int KM = 8>
__global__ __launch_bounds__(64 * CALL_WG_CHAINS, 2) void call_mcmc_kernel(const CallParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  // (wave-uniform by construction; said so, or everything derived from it -- the chain, its pointers -- lives in vector registers)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);
  const int unit = blockIdx.y, chain = (int)blockIdx.x * nwv + wv;
  const int lane = (int)(threadIdx.x & 63);
  const int R = P.R, H = P.H, K = P.K;
  ExactParams EP;
  EP.reads = P.reads;
  EP.counts = P.counts;
  EP.haps = P.haps;
  EP.inbreeding = P.inbreeding;
  EP.freqs = P.freqs;
  EP.R = R; EP.M = P.M; EP.A = P.A; EP.H = H; EP.K = K;
  EP.has_prior = P.has_prior;
  EP.ptab_scale = 1.0;
  EP.Rcap = 0;  // the whole product table (the sampler does not tile the reads)
  // (a table in the workspace is a table per chain: the host then launches one chain per workgroup)
  EP.ptab_ext = P.ptab_ext ? P.ptab_ext + ((size_t)unit * P.chains + chain) * ((size_t)R * H + R) : nullptr;
  if (P.phase == 1) {  // resume launch: a workgroup none of whose chains was handed back has nothing to do (uniform over the workgroup)
    bool any = false;
    for (int c = (int)blockIdx.x * nwv; c < ((int)blockIdx.x + 1) * nwv && c < P.chains; c++)
      any |= (unsigned)(P.state[((size_t)unit * P.chains + c) * P.state_stride] >> 32) == CALL_RESUME;
    if (!any) return;
  }
  ExactLds E;
  PriorTab pt;
  exact_setup(EP, unit, smem, E, pt);       // (all wavefronts of the workgroup together)
  double *rtab = E.red;                     // [H][K]: lgamma(1 + alpha_a + ibs) - lgamma(alpha_a + ibs)   (shared)
  double *o_llk = rtab + (size_t)H * K + (size_t)wv * call_lds_private(H);   // [H] each: this chain's
  double *o_lpr = o_llk + H;
  double *o_prob = o_lpr + H;
  double *o_aux = o_prob + H;               // proposal ratios (Metropolis-Hastings)
  // Gibbs memo (call_memo_entries): keys (context rank + 1, 0 = empty), then per entry H probabilities and H likelihoods
  // (the memo's key packs the K - 1 other alleles eight bits each: ploidies above 8 -- round 5 -- do without the memo)
  const int n_memo = (P.step_type == 0 && 8 * (K - 1) <= 63) ? call_memo_entries(H) : 0;
  long long *memo_key = reinterpret_cast<long long *>(o_aux + H + 64);
  double *memo_val = reinterpret_cast<double *>(memo_key + CALL_MEMO);
  uint64_t *state = P.state ? P.state + ((size_t)unit * P.chains + (chain < P.chains ? chain : 0)) * P.state_stride : nullptr;
  const bool resumed = P.phase == 1;
  if (!resumed)
    for (int i = lane; i < CALL_MEMO; i += WAVE) memo_key[i] = 0;
  int memo_next = 0;                        // (wave-uniform) the entry the next miss replaces
  __shared__ double s_acc_[CALL_WG_CHAINS], s_choice_llk_[CALL_WG_CHAINS];
  __shared__ int s_g_[CALL_WG_CHAINS][KM];     // the chain's genotype (array order)
  __shared__ int s_req_[CALL_WG_CHAINS][KM];   // alleles of the request being evaluated
  __shared__ double s_left;                 // Gibbs prior: lgamma(sum_alpha) - lgamma(1 + sum_alpha)   (shared)
  __shared__ int s_choice_[CALL_WG_CHAINS];
  __shared__ int s_full_[CALL_WG_CHAINS];
  // (LDS-qualified: through a generic pointer picked by the wavefront's index these become flat loads -- measured 150 -> 194 ms)
  typedef __attribute__((address_space(3))) double lds_f64;
  typedef __attribute__((address_space(3))) int lds_i32;
  lds_f64 &s_acc = *(lds_f64 *)&s_acc_[wv], &s_choice_llk = *(lds_f64 *)&s_choice_llk_[wv];
  lds_i32 *s_g = (lds_i32 *)s_g_[wv], *s_req = (lds_i32 *)s_req_[wv];
  lds_i32 &s_choice = *(lds_i32 *)&s_choice_[wv], &s_full = *(lds_i32 *)&s_full_[wv];
  const bool has_prior = P.has_prior != 0;
  const bool has_freqs = has_prior && P.freqs != nullptr;
  const double F = pt.F;
  if (lane == 0) s_full = 0;
  if (has_prior && F != 0.0) {
    const double scale = (1.0 - F) / F;
    for (int q = (int)threadIdx.x; q < H * K; q += (int)blockDim.x) {
      const int a = q / K, ibs = q % K;
      const double alpha = has_freqs ? P.freqs[(size_t)unit * H + a] * scale : (1.0 / (double)H) * scale;
      const double va = alpha + (double)ibs;
      rtab[q] = lgamma(1.0 + va) - lgamma(va);
    }
    if (threadIdx.x == 0) {
      double sum_alpha;
      if (has_freqs) {
        double s = 0.0;
        for (int h = 0; h < H; h++) s += P.freqs[(size_t)unit * H + h] * scale;
        sum_alpha = (double)(K - 1) + s;
      } else {
        sum_alpha = (double)(K - 1) + ((1.0 / (double)H) * scale) * (double)H;
      }
      s_left = lgamma(sum_alpha) - lgamma(1.0 + sum_alpha);
    }
  }
  __syncthreads();  // (the last workgroup-wide barrier: from here on every wavefront runs its chain alone)
  if (chain >= P.chains) return;
  int first_step = 0;
  if (resumed) {
    const uint64_t w0 = state[0];
    if ((unsigned)(w0 >> 32) != CALL_RESUME) return;
    first_step = __builtin_amdgcn_readfirstlane((int)(unsigned)w0);
    memo_next = __builtin_amdgcn_readfirstlane((int)state[1]);
    const int nw = CALL_MEMO + n_memo * 2 * H;  // keys and entries: one run of words in the record and in LDS
    uint64_t *dst = reinterpret_cast<uint64_t *>(memo_key);
    for (int i = lane; i < nw; i += WAVE) dst[i] = state[CALL_STATE_HDR + i];
  }
  const double invK_full = 1.0 / (double)K;
  ulonglong2 *cache = P.cache + ((size_t)unit * P.chains + chain) * (size_t)P.cache_slots;
  const unsigned long long cmask = (unsigned long long)P.cache_slots - 1ull;

  // likelihood of the request in s_req (ploidy kk <= K), whole wave: lanes over reads, sums in read order per lane,
  // then the wave butterfly (assemble/likelihood.py:17-70 up to the order of the sum over reads)
  auto coop_llk = [&](int kk) -> double {
    const double invk = 1.0 / (double)kk;
    double s = 0.0;
    if (EP.ptab_ext) {  // table in the workspace
      for (int r = lane; r < R; r += WAVE) {
        const double *row = E.ptab + (size_t)r * H;
        double rp = 0.0;
        for (int i = 0; i < kk; i++) rp += row[s_req[i]] * invk;
        s += read_log(rp) * E.cnt[r];
      }
    } else {  // table in the LDS: ds_read instead of flat_load (exact_kernel.hpp lds_table), same values
      lds_cdouble *ptab = lds_table(E.ptab), *cnt = lds_table(E.cnt);
      int r = lane;
      if (E.w01) {
        // unweighted reads (round 5): the lane's reads four at a time, ONE logarithm for their product (read_log_product) --
        // a lane without a fourth read takes the factor 1
        for (; r < R; r += 4 * WAVE) {
          double rp[4];
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int rr = r + t * WAVE;
            rp[t] = 0.0;
            if (rr < R) {
              lds_cdouble *row = ptab + rr * H;
              for (int i = 0; i < kk; i++) rp[t] += row[s_req[i]] * invk;
            } else {
              rp[t] = 1.0;
            }
          }
          s += read_log_product<4>(rp);
        }
      } else {
        for (; r < R; r += WAVE) {
          lds_cdouble *row = ptab + r * H;
          double rp = 0.0;
          for (int i = 0; i < kk; i++) rp += row[s_req[i]] * invk;
          s += read_log(rp) * cnt[r];
        }
      }
    }
    return wave_sum(s);
  };
  (void)invK_full;

  // ---- initial genotype: the handed-back chain's, the caller's, or greedy_caller (calling/mcmc.py:393-453) ----
  if (resumed) {
    if (lane < K) s_g[lane] = (int)reinterpret_cast<const uint32_t *>(state + 2)[lane];
    call_sync();
  } else if (P.initial) {
    if (lane < K) s_g[lane] = (int)P.initial[(size_t)unit * K + lane];
    call_sync();
  } else {
    for (int i = 0; i < K; i++) {
      double best = -INFINITY;
      int best_a = -1;
      for (int a = 0; a < H; a++) {
        if (lane == 0) {
          for (int q = 0; q < i; q++) s_req[q] = s_g[q];
          s_req[i] = a;
        }
        call_sync();
        const double llk = coop_llk(i + 1);
        double lprior = 0.0;
        if (has_prior) {
          // the prior of a genotype of ploidy i + 1: its tables depend on the ploidy (left term, lgamma(ploidy + 1))
          int g[KM];
          for (int q = 0; q <= i; q++) g[q] = s_req[q];
          const int kk = i + 1;
          if (F == 0.0) {
            PriorTab t2 = pt;
            double den = 0.0;
            for (int x = 0; x < kk; x++) {
              bool first = true;
              int dose = 0;
              for (int y = 0; y < kk; y++)
                if (g[y] == g[x]) {
                  dose++;
                  if (y < x) first = false;
                }
              den += first ? lgamma((double)dose + 1.0) : 0.0;
            }
            const double ln_perms = lgamma((double)kk + 1.0) - den;
            if (!t2.has_freqs) lprior = ln_perms - (double)kk * t2.lnH;
            else {
              double prod = 1.0;
              for (int q = 0; q < kk; q++) prod *= t2.lfreq[g[q]];
              lprior = ln_perms + log(prod);
            }
          } else {
            const double scale = (1.0 - F) / F;
            double sum_alphas;
            if (has_freqs) {
              sum_alphas = 0.0;
              for (int h = 0; h < H; h++) sum_alphas += P.freqs[(size_t)unit * H + h] * scale;
            } else {
              sum_alphas = ((1.0 / (double)H) * scale) * (double)H;
            }
            const double left = (lgamma((double)kk + 1.0) + lgamma(sum_alphas)) - lgamma((double)kk + sum_alphas);
            double prod = 0.0;
            for (int x = 0; x < kk; x++) {
              bool first = true;
              int dose = 0;
              for (int y = 0; y < kk; y++)
                if (g[y] == g[x]) {
                  dose++;
                  if (y < x) first = false;
                }
              if (first) {
                const double alpha = has_freqs ? P.freqs[(size_t)unit * H + g[x]] * scale : (1.0 / (double)H) * scale;
                prod += lgamma((double)dose + alpha) - (lgamma((double)dose + 1.0) + lgamma(alpha));
              }
            }
            lprior = left + prod;
          }
        }
        const double lprob = llk + lprior;
        if (lprob > best) {
          best = lprob;
          best_a = a;
        }
        call_sync();
      }
      if (lane == 0) s_g[i] = best_a;
      call_sync();
    }
    // genotype.sort()
    if (lane == 0) {
      for (int a = 1; a < K; a++) {
        const int v = s_g[a];
        int b = a - 1;
        while (b >= 0 && s_g[b] > v) {
          s_g[b + 1] = s_g[b];
          b--;
        }
        s_g[b + 1] = v;
      }
    }
    call_sync();
  }

  CallStream st;
  st.k0 = (uint32_t)P.seed;
  st.k1 = (uint32_t)(P.seed >> 32) ^ (uint32_t)(P.stream_ids[unit] >> 32);
  st.c2 = ((uint32_t)chain << 16) | SLOT_CALL;
  st.c3 = (uint32_t)P.stream_ids[unit];
  uint64_t ctr = (uint64_t)first_step * (uint64_t)(2 * K - 1);  // (K - 1 shuffle draws and K uniforms per step)
  int64_t *gout = P.genotypes + (((size_t)unit * P.chains + chain) * P.steps) * K;
  double *lout = P.llks + ((size_t)unit * P.chains + chain) * P.steps;

  // probe the chain's table for `key`: returns true and the value, or false and the slot to fill
  auto probe = [&](long long key, double &val, ulonglong2 *&slot) -> bool {
    unsigned long long h = (unsigned long long)key * 0x9E3779B97F4A7C15ull;
    unsigned long long i = (h >> 20) & cmask;
    for (long long tries = 0; tries < P.cache_slots; tries++) {
      ulonglong2 e = cache[i];
      if (e.x == (unsigned long long)key + 1ull) {
        val = __longlong_as_double((long long)e.y);
        return true;
      }
      if (e.x == 0ull) {
        slot = cache + i;
        return false;
      }
      i = (i + 1) & cmask;
    }
    slot = nullptr;  // full
    return false;
  };

  // llks of the options `a` in [a0, a0 + 64) of a sub-step at position k, into o_llk: cache probe per lane, misses one by one
  auto option_llks = [&](int k, int a0) {
    const int a = a0 + lane;
    const bool act = a < H;
    int g[KM];
    for (int i = 0; i < K; i++) g[i] = s_g[i];
    g[k] = act ? a : g[k];
    double val = 0.0;
    ulonglong2 *slot = nullptr;
    bool miss = false;
    long long key = 0;
    if (act) {
      key = call_key(g, K);
      miss = !probe(key, val, slot);
      if (miss && !slot) s_full = 1;
    }
    unsigned long long todo = __ballot(miss);
    const bool any_miss = todo != 0ull;
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      if (lane == src)
        for (int i = 0; i < K; i++) s_req[i] = g[i];
      call_sync();
      const double v = coop_llk(K);
      if (lane == src) val = v;
    }
    // The lanes' new entries go into the table together (round 5; until then one after the other, a probe and a fence over global
    // memory each: 4 of a miss's 8 us): a lane claims the first empty slot of its probe sequence by compare-and-swap on the key
    // word, then stores the value -- two lanes that end on the same empty slot are told apart by the swap, and nobody reads the
    // table before the fence below.  Which slot a key lands in depends on the race; its value does not (the keys of a round
    // differ: the options differ in the allele at k).  The table never forgets an entry -- the reference's dict keeps the value of
    // whichever allele order was evaluated first (calling/likelihood.py:36-78).
    if (miss && slot) {
      unsigned long long h = (unsigned long long)key * 0x9E3779B97F4A7C15ull;
      unsigned long long i = (h >> 20) & cmask;
      bool placed = false;
      for (long long tries = 0; tries < P.cache_slots && !placed; tries++) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&cache[i].x), 0ull, (unsigned long long)key + 1ull);
        if (old == 0ull) {
          cache[i].y = (unsigned long long)__double_as_longlong(val);
          placed = true;
        } else if (old == (unsigned long long)key + 1ull) {
          placed = true;
        } else {
          i = (i + 1) & cmask;
        }
      }
      if (!placed) s_full = 1;
    }
    if (any_miss) call_sync_global();  // (the next sub-step's probes read behind these entries)
    if (act) o_llk[a] = val;
  };

  const bool hand_over = state != nullptr && !P.last && n_memo > 0;
  bool handed = false;
  for (int step = first_step; step < P.steps; step++) {
    bool all_known = true;  // every sub-step of this step found its context in the memo
    // np.random.shuffle(arange(ploidy)) -- every lane the same
    int order[KM];
    for (int i = 0; i < K; i++) order[i] = i;
    for (int i = K - 1; i >= 1; i--) {
      const int j = (int)call_interval(st, ctr++, (uint32_t)i);
      const int t = order[i];
      order[i] = order[j];
      order[j] = t;
    }
    int choice = 0;
    for (int jj = 0; jj < K; jj++) {
      const int k = order[jj];
      const int current = s_g[k];
      double cur_llk = 0.0, cur_lprior = 0.0;
      int cur_copies = 1;
      if (P.step_type == 1) {
        // mh_options (calling/mcmc.py:15-140): likelihood and prior of the current genotype first
        int g[KM];
        for (int i = 0; i < K; i++) g[i] = s_g[i];
        cur_copies = 0;
        for (int i = 0; i < K; i++) cur_copies += g[i] == current ? 1 : 0;
        if (has_prior) cur_lprior = calling_log_prior_unsorted(pt, g, K);
        double val = 0.0;
        ulonglong2 *slot = nullptr;
        const long long key = call_key(g, K);
        bool hit = false;
        if (lane == 0) {
          hit = probe(key, val, slot);
          if (!hit && !slot) s_full = 1;
        }
        hit = __shfl((int)hit, 0, WAVE) != 0;
        if (!hit) {
          if (lane == 0)
            for (int i = 0; i < K; i++) s_req[i] = g[i];
          call_sync();
          val = coop_llk(K);
          if (lane == 0 && slot) *slot = make_ulonglong2((unsigned long long)key + 1ull, (unsigned long long)__double_as_longlong(val));
          call_sync_global();
        }
        cur_llk = __shfl(val, 0, WAVE);
      }
      // Gibbs: has this context (the other K - 1 alleles) been normalised before?
      const double *use_prob = o_prob, *use_llk = o_llk;
      long long ctx = 0;
      int memo_hit = -1;
      if (n_memo > 0) {
        // key of the context: up to sixteen known haplotypes, the copies of each among the other alleles, four bits a haplotype
        // (call_ctx_counts); more: the other alleles sorted, eight bits each (H <= 256, K - 1 <= 7) -- any one-to-one function of
        // the multiset does; the genotype's rank would cost a chain of 64-bit divisions per sub-step.  Sorted by counting
        // (the place of an allele is the number of alleles before it in the order), all loops of constant extent: no
        // register array is indexed by a run-time value
        if (H <= 16 && K >= 2) {
          ctx = (long long)call_ctx_counts<KM>([&](int i) { return s_g[i]; }, K, k);
        } else {
        int v[KM];
#pragma unroll
        for (int i = 0; i < KM; i++) v[i] = i < K ? s_g[i] : 0;
        unsigned long long packed = 0ull;
#pragma unroll
        for (int i = 0; i < KM; i++) {
          int place = 0;
#pragma unroll
          for (int j = 0; j < KM; j++)
            place += (j < K && j != k && j != i && (v[j] < v[i] || (v[j] == v[i] && j < i))) ? 1 : 0;
          if (i < K && i != k) packed |= (unsigned long long)(v[i] & 255) << (8 * place);
        }
        ctx = (long long)((packed << 1) | 1ull);  // (never 0: 0 marks an empty entry)
        }
        const unsigned long long m = __ballot(lane < n_memo && memo_key[lane] == ctx);
        memo_hit = m ? __ffsll((long long)m) - 1 : -1;
      }
      if (memo_hit >= 0) {
        use_prob = memo_val + (size_t)memo_hit * 2 * H;
        use_llk = use_prob + H;
      } else {
      all_known = false;
      for (int a0 = 0; a0 < H; a0 += WAVE) option_llks(k, a0);
      // priors (and proposal ratios) of the options
      for (int a = lane; a < H; a += WAVE) {
        int g[KM];
        for (int i = 0; i < K; i++) g[i] = s_g[i];
        g[k] = a;
        int copies = 0;
        for (int i = 0; i < K; i++) copies += g[i] == a ? 1 : 0;
        if (P.step_type == 0) {
          double lp;
          if (!has_prior) lp = log((double)copies);  // log_genotype_allele_flat_prior (prior.py:30-52)
          else if (F == 0.0) lp = has_freqs ? log(pt.lfreq[a]) : log(1.0 / (double)H);
          else lp = s_left + rtab[a * K + (copies - 1)];
          o_lpr[a] = lp;
        } else {
          if (a == current) {
            o_lpr[a] = cur_lprior;
            o_llk[a] = cur_llk;
            o_aux[a] = 0.0;
          } else {
            o_lpr[a] = has_prior ? calling_log_prior_unsorted(pt, g, K) : 0.0;
            o_aux[a] = log((double)copies / (double)cur_copies);
          }
        }
      }
      call_sync();
      if (P.step_type == 0) {
        // normalise_log_probs(llks + lpriors): sequential add_log_prob in allele order (jitutils.py:30-74) -- one lane --,
        // then the H exponentials, one lane each (the same function of the same arguments as the sequential loop)
        if (lane == 0) {
          double acc = o_llk[0] + o_lpr[0];
          for (int a = 1; a < H; a++) acc = add_log_prob(acc, o_llk[a] + o_lpr[a]);
          s_acc = acc;
        }
        call_sync();
        const double acc = s_acc;
        for (int a = lane; a < H; a += WAVE) o_prob[a] = exp((o_llk[a] + o_lpr[a]) - acc);
        call_sync();
        if (n_memo > 0) {  // remember the context
          double *mv = memo_val + (size_t)memo_next * 2 * H;
          for (int a = lane; a < H; a += WAVE) {
            mv[a] = o_prob[a];
            mv[H + a] = o_llk[a];
          }
          if (lane == 0) memo_key[memo_next] = ctx;
          memo_next = memo_next + 1 == n_memo ? 0 : memo_next + 1;
          call_sync();
        }
      }
      }  // (memo miss)
      if (P.step_type != 0) {
        if (lane == 0) {
          double sum = 0.0;
          for (int a = 0; a < H; a++) {
            const double r = ((o_llk[a] - cur_llk) + (o_lpr[a] - cur_lprior)) + o_aux[a];
            o_prob[a] = exp(fmin(0.0, r));
          }
          o_prob[current] = 0.0;
          for (int a = 0; a < H; a++) o_prob[a] /= (double)(H - 1);
          for (int a = 0; a < H; a++) sum += o_prob[a];
          o_prob[current] = 1.0 - sum;
        }
        call_sync();
      }
      // random_choice: searchsorted(cumsum(p), u, side="right") -- lane 0 walks the cumulative sum and stops at the choice (a settled
      // chain's mass sits on the first few alleles: two or three reads; forming all H sums in every lane by v_readlane was
      // measured slower, round 5: 118.9 -> 134.2 ms)
      if (lane == 0) {
        const double u = call_double(st, ctr);
        double cacc = 0.0;
        int ch = H;
        for (int a = 0; a < H; a++) {
          cacc += use_prob[a];
          if (cacc > u) {
            ch = a;
            break;
          }
        }
        if (ch >= H) ch = H - 1;  // u beyond the last cumulative value (probability ~1e-16)
        s_choice = ch;
        s_choice_llk = use_llk[ch];
        s_g[k] = ch;
      }
      ctr++;
      call_sync();
      choice = s_choice;
    }
    // genotype_alleles.sort(); the step's llk is that of the last choice
    // (tried in round 5 and not kept: the sort by counting, one lane per allele -- 118.9 -> 121.5 ms; the serial part of a sub-step
    // is not where its time goes)
    if (lane == 0) {
      for (int a = 1; a < K; a++) {
        const int v = s_g[a];
        int b = a - 1;
        while (b >= 0 && s_g[b] > v) {
          s_g[b + 1] = s_g[b];
          b--;
        }
        s_g[b + 1] = v;
      }
      lout[step] = s_choice_llk;
    }
    call_sync();
    if (lane < K) gout[(size_t)step * K + lane] = s_g[lane];
    call_sync();
    // A step that met only remembered contexts: the chain has settled -- its record to the workspace, the rest of its steps to
    // call_coast_kernel (a lane per chain; it hands the chain back at the first context it does not know)
    if (hand_over && all_known && step + 1 < P.steps) {
      const int nw = CALL_MEMO + n_memo * 2 * H;
      const uint64_t *src = reinterpret_cast<const uint64_t *>(memo_key);
      for (int i = lane; i < nw; i += WAVE) state[CALL_STATE_HDR + i] = src[i];
      if (lane < 8) reinterpret_cast<uint32_t *>(state + 2)[lane] = lane < K ? (uint32_t)s_g[lane] : 0u;
      if (lane == 0) {
        state[1] = (uint64_t)memo_next;
        state[0] = (uint64_t)(unsigned)(step + 1) | ((uint64_t)CALL_COAST << 32);
      }
      handed = true;
      break;
    }
  }
  if (state && !handed && lane == 0) state[0] = (uint64_t)(unsigned)P.steps | ((uint64_t)CALL_DONE << 32);
  if (lane == 0 && s_full) atomicMin(&P.status[unit], MCHAP_ERR_LIMIT);
}

// ---- settled chains: a lane per chain ----
// A chain of `mchap call` that has settled meets the same K contexts step after step, and a sub-step in a remembered context is a
// table look-up and a draw: nothing in it needs a wavefront.  call_mcmc_kernel hands such a chain over after its first step of
// remembered contexts only (the record of call_state_words); here `chains_per_wave` chains run side by side, one per lane, their
// memos in LDS, each until its last step or the first context its memo does not hold -- that step is left undone and the chain
// goes back to call_mcmc_kernel (phase 1), which finds the draws of a step by its number.  The same draws (Philox counters by
// step), the same probabilities (the memo's doubles), the same sequential cumulative sum: the same traces.

// sorting network of eight (19 compare-exchanges)
__device__ __forceinline__ void call_cx(unsigned &a, unsigned &b) {
  const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}
__device__ __forceinline__ void call_sort8(unsigned (&v)[8]) {
  call_cx(v[0], v[1]); call_cx(v[2], v[3]); call_cx(v[4], v[5]); call_cx(v[6], v[7]);
  call_cx(v[0], v[2]); call_cx(v[1], v[3]); call_cx(v[4], v[6]); call_cx(v[5], v[7]);
  call_cx(v[1], v[2]); call_cx(v[5], v[6]); call_cx(v[0], v[4]); call_cx(v[3], v[7]);
  call_cx(v[1], v[5]); call_cx(v[2], v[6]);
  call_cx(v[1], v[4]); call_cx(v[3], v[6]);
  call_cx(v[2], v[4]); call_cx(v[3], v[5]);
  call_cx(v[3], v[4]);
}
// Philox draws of a lane's stream with the last block kept (two draws a block: philox.hpp)
struct CallDraws {
  CallStream s;
  uint64_t blk;
  uint32_t o[4];
  __device__ __forceinline__ void words(uint64_t n, uint32_t &a, uint32_t &b) {
    if ((n >> 1) != blk) {
      blk = n >> 1;
      philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s.c2, s.c3, s.k0, s.k1, o);
    }
    a = (n & 1) ? o[2] : o[0];
    b = (n & 1) ? o[3] : o[1];
  }
};

// Lane l of wavefront w (CALL_COAST_WAVES of them a workgroup: one per SIMD of a compute unit) runs chain w * chains_per_wave + l
// (unit-major, as the records lie); lds_stride: 8-byte words of a chain's memo in LDS (odd: the lanes' accesses spread over the
// banks).  A step is a serial program of ~1300 instructions whatever the number of lanes in use, so a batch is spread as thin as
// the chip allows: few lanes per wavefront, a wavefront per SIMD (the host picks chains_per_wave).
constexpr int CALL_COAST_WAVES = 4;
__global__ __launch_bounds__(64 * CALL_COAST_WAVES) void call_coast_kernel(const CallParams P, const int chains_per_wave, const int lds_stride) {
  extern __shared__ __align__(16) unsigned char smem[];
  typedef __attribute__((address_space(3))) uint64_t lds_u64;
  typedef __attribute__((address_space(3))) const double lds_f64c;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  lds_u64 *cm = (lds_u64 *)reinterpret_cast<uint64_t *>(smem) + wv * chains_per_wave * lds_stride;
  const int lane = (int)(threadIdx.x & 63);
  const int H = P.H, K = P.K;
  const int n_memo = call_memo_entries(H);
  const int nw = CALL_MEMO + n_memo * 2 * H;
  const long long n_all = (long long)P.n_units * P.chains;
  const long long c0 = ((long long)blockIdx.x * CALL_COAST_WAVES + wv) * chains_per_wave;
  const long long ci = c0 + lane;
  bool live = lane < chains_per_wave && ci < n_all;
  uint64_t *state = P.state + (size_t)(live ? ci : 0) * P.state_stride;
  int step = 0;
  if (live) {
    const uint64_t w0 = state[0];
    live = (unsigned)(w0 >> 32) == CALL_COAST;
    step = (int)(unsigned)w0;
  }
  const unsigned long long mine = __ballot(live);
  if (!mine) return;
  // the memos of this wavefront's chains into LDS, a chain at a time (the record's words are contiguous)
  for (int l = 0; l < chains_per_wave; l++) {
    if (!((mine >> l) & 1ull)) continue;
    const uint64_t *src = P.state + (size_t)(c0 + l) * P.state_stride + CALL_STATE_HDR;
    for (int i = lane; i < nw; i += WAVE) cm[l * lds_stride + i] = src[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  lds_u64 *keys = cm + lane * lds_stride;
  lds_f64c *vals = (lds_f64c *)(keys + CALL_MEMO);
  // ... and each entry's probabilities turned into their running sums, added in allele order as random_choice's cumsum adds them:
  // the draw is then the number of sums not above u (they never decrease) -- a search instead of a walk
  if (live) {
    typedef __attribute__((address_space(3))) double lds_f64;
    for (int e = 0; e < n_memo; e++) {
      lds_f64 *pr = (lds_f64 *)(keys + CALL_MEMO) + (size_t)e * 2 * H;
      double acc = 0.0;
      for (int x = 0; x < H; x++) {
        acc += pr[x];
        pr[x] = acc;
      }
    }
  }
  // the genotype: eight alleles of eight bits (the memo's key holds K - 1 <= 7 of them: K <= 8, H <= 256)
  uint64_t gp = 0;
  if (live)
    for (int i = 0; i < 8; i++) gp |= (uint64_t)(reinterpret_cast<const uint32_t *>(state + 2)[i] & 255u) << (8 * i);
  const bool few = H <= 16 && K >= 2;
  // (few: the copies of each haplotype in the genotype, four bits each -- call_ctx_counts of a context is this less one allele)
  uint64_t counts = 0;
  if (few)
    for (int i = 0; i < K; i++) counts += 1ull << (4 * (int)((gp >> (8 * i)) & 15ull));
  const int unit = live ? (int)(ci / P.chains) : 0, chain = live ? (int)(ci % P.chains) : 0;
  CallDraws D;
  D.s.k0 = (uint32_t)P.seed;
  D.s.k1 = (uint32_t)(P.seed >> 32) ^ (uint32_t)(P.stream_ids[unit] >> 32);
  D.s.c2 = ((uint32_t)chain << 16) | SLOT_CALL;
  D.s.c3 = (uint32_t)P.stream_ids[unit];
  D.blk = ~0ull;
  int64_t *gout = P.genotypes + ((size_t)(live ? ci : 0) * P.steps) * K;
  double *lout = P.llks + (size_t)(live ? ci : 0) * P.steps;
  const int per_step = 2 * K - 1;
  int search_rounds = 0;  // of the binary search over H sums
  while ((1 << search_rounds) <= H) search_rounds++;

  while (__ballot(live)) {
    if (live) {
      uint64_t ctr = (uint64_t)step * (uint64_t)per_step;
      // np.random.shuffle(arange(ploidy)): positions four bits each
      uint32_t order = 0x76543210u;
      for (int i = K - 1; i >= 1; i--) {
        uint32_t a, b;
        D.words(ctr++, a, b);
        const int j = (int)__umulhi(a, (uint32_t)i + 1u);
        const uint32_t oi = (order >> (4 * i)) & 15u, oj = (order >> (4 * j)) & 15u;
        order = (order & ~(15u << (4 * i))) | (oj << (4 * i));
        order = (order & ~(15u << (4 * j))) | (oi << (4 * j));
      }
      uint64_t g = gp, cn = counts;
      double choice_llk = 0.0;
      lds_f64c *llk_at = vals;
      bool known = true;
      for (int jj = 0; jj < K; jj++) {
        const int k = (int)((order >> (4 * jj)) & 15u);
        // the context's key as call_mcmc_kernel forms it
        uint64_t ctx;
        if (few) {
          ctx = cn - (1ull << (4 * (int)((g >> (8 * k)) & 15ull)));
        } else {
          unsigned v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) v[i] = (i < K && i != k) ? (unsigned)((g >> (8 * i)) & 255ull) : 0x1FFu;
          call_sort8(v);
          unsigned long long packed = 0ull;
#pragma unroll
          for (int i = 0; i < 7; i++)
            if (i < K - 1) packed |= (unsigned long long)v[i] << (8 * i);
          ctx = (packed << 1) | 1ull;
        }
        int hit = -1;
#pragma unroll
        for (int e = 0; e < CALL_MEMO; e++)  // (entries the memo does not use keep the key 0)
          if (keys[e] == ctx) hit = e;
        if (hit < 0) {
          known = false;
          break;
        }
        lds_f64c *pr = vals + (size_t)hit * 2 * H;
        uint32_t a, b;
        D.words(ctr++, a, b);
        const double u = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
        // random_choice: searchsorted(cumsum(p), u, side="right") = how many of the running sums are not above u (beyond the
        // last: the last allele, as call_mcmc_kernel)
        int ch = 0;
        if (few) {
          double q[16];
#pragma unroll
          for (int x = 0; x < 16; x++) q[x] = pr[x < H ? x : H - 1];
#pragma unroll
          for (int x = 0; x < 16; x++) ch += (x < H && !(q[x] > u)) ? 1 : 0;
        } else {
          int lo = 0, hi = H;  // the first sum above u lies in [lo, hi]
          for (int r = 0; r < search_rounds; r++) {
            const int mid = (lo + hi) >> 1;
            const double c = pr[mid < H ? mid : H - 1];
            if (lo < hi) {
              if (c > u) hi = mid;
              else lo = mid + 1;
            }
          }
          ch = lo;
        }
        if (ch > H - 1) ch = H - 1;
        llk_at = pr + H + ch;  // (the step's llk is that of its last choice: read once, after the sub-steps)
        g = (g & ~(255ull << (8 * k))) | ((uint64_t)ch << (8 * k));
        if (few) cn = ctx + (1ull << (4 * ch));
      }
      if (known) choice_llk = *llk_at;
      if (!known) {
        // this step is call_mcmc_kernel's: the record keeps the genotype of the step's start (the memo is as it was)
        for (int i = 0; i < 8; i++) reinterpret_cast<uint32_t *>(state + 2)[i] = (uint32_t)((gp >> (8 * i)) & 255ull);
        state[0] = (uint64_t)(unsigned)step | ((uint64_t)CALL_RESUME << 32);
        live = false;
      } else {
        // genotype_alleles.sort(); the step's llk is that of the last choice
        unsigned v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = i < K ? (unsigned)((g >> (8 * i)) & 255ull) : 0x1FFu;
        call_sort8(v);
        gp = 0;
#pragma unroll
        for (int i = 0; i < 8; i++)
          if (i < K) {
            gp |= (uint64_t)v[i] << (8 * i);
            gout[(size_t)step * K + i] = (int64_t)v[i];
          }
        lout[step] = choice_llk;
        counts = cn;
        step++;
        if (step >= P.steps) {
          state[0] = (uint64_t)(unsigned)P.steps | ((uint64_t)CALL_DONE << 32);
          live = false;
        }
      }
    }
  }
}

}  // namespace mchap
