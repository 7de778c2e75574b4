// Trace -> posterior summary on the device: replaces the host-side Python of the reference's
// GenotypeMultiTrace.burn(n).posterior() (assemble/classes.py:280-325, mset.unique_counts mset.py:361-392)
// and PosteriorGenotypeDistribution.mode_genotype_support() / GenotypeSupportDistribution.mode_genotype()
// (assemble/classes.py:87-128,194-205) as read by application/assemble.py:144-157.
//
// One wavefront per unit.  States (sorted genotypes = K packed haplotype words, written by the sampler) are
// visited in the reference's merged order (chain-major, steps ascending after burn-in), 64 at a time; a list of
// distinct states is kept in LDS in order of first appearance with their counts, exactly the order
// mset.unique_counts produces.  The list is then ranked by (count descending, first appearance descending):
// the order np.flip(np.argsort(probs)) gives for tied probabilities.
#pragma once
#include <limits.h>

#include "denovo_kernel.hpp"

namespace mchap {

constexpr int POST_CAP = 512;  // distinct states kept per unit by the batch launches; more -> overflow flag (post_n = -n), and the
                               // few units that overflow are summarised again by a listed launch with a larger table (cap)

struct PosteriorParams {
  const mchap_unit *units;
  const uint64_t *trace;
  int steps, chains, burn;
  int max_states, ploidy_max;
  int wph;                   // 64-bit words per haplotype of the traces: 1, or 2 for a batch of the general sampler (units of more than
                             // 64 bits per haplotype, ploidies 9 to 15): a state is ploidy x wph words, and so are the rows of
                             // post_words / mode_words (ploidy_max x wph words each)
  int cap;                   // distinct states the LDS table holds (POST_CAP for a whole batch; up to posterior_max_cap for a list)
  const int32_t *unit_list;  // null: workgroup b summarises unit b; else unit unit_list[b], its states written to row b
  uint64_t *post_words;
  int32_t *post_counts;
  int32_t *post_n;
  double *mode_stats;
  int32_t *mode_index;
  uint64_t *mode_words;   // [U][ploidy_max] or null: the mode genotype itself (its rank may exceed max_states)
  int32_t *mode_count;    // [U] or null
};

// Distinct states of chains [ch_lo, ch_hi) after burn-in, ranked, with support labels and the mode support.
// LDS: uw [POST_CAP][K * W], ucount / order / label [POST_CAP].  Every lane returns the same n_u, overflow, best, best_r.
// SW: the bound of a state's words (8: the fast samplers' traces; 32: ploidies up to 15 at W = 2 words per haplotype -- round 5)
struct PostSummary {
  int n_u, overflow, best_r;
  double best;  // probability of the mode support (SPM)
};

// slots of the hash index beside a table of `cap` states: the power of two at or above 2 cap
__host__ __device__ inline int post_hash_slots(int cap) {
  int hs = 64;
  while (hs < 2 * cap) hs <<= 1;
  return hs;
}
__device__ __forceinline__ bool post_hap_eq(const uint64_t *a, const uint64_t *b, int W) {
  bool eq = a[0] == b[0];
  if (W > 1) eq = eq && a[1] == b[1];
  return eq;
}
template <int SW>
__device__ __forceinline__ PostSummary post_summarise(const uint64_t *trace, const mchap_unit &U, int S, int burn, int ch_lo,
                                                      int ch_hi, uint64_t *uw, int *ucount, int *order, int *label, const int CAP,
                                                      const int W, int *hidx, const int HS) {
  const int Kh = U.ploidy;   // haplotypes of a state
  const int K = Kh * W;      // words of a state
  // hidx [HS] (HS a power of two >= 2 CAP): open-addressing index of the distinct states by a hash of their words -- a state of
  // the trace is matched by a probe instead of a walk over everything seen so far (round 5: the walk was quadratic in the distinct
  // states, 20 ms for docs/example's 24 wandering units)
  for (int i = (int)threadIdx.x; i < HS; i += WAVE) hidx[i] = -1;
  const int lane = threadIdx.x;
  const int per_chain = S - burn;
  const int N = (ch_hi - ch_lo) * per_chain;
  for (int i = lane; i < CAP; i += WAVE) ucount[i] = 0;
  __syncthreads();
  const unsigned hmask = (unsigned)HS - 1u;

  int n_u = 0;        // wave-uniform
  int overflow = 0;
  for (int base = 0; base < N; base += WAVE) {
    const int n = base + lane;
    const bool active = n < N;
    uint64_t st[SW];
    if (active) {
      const int ch = ch_lo + n / per_chain, s = burn + n % per_chain;
      const uint64_t *src = trace + U.trace_off + ((size_t)ch * S + s) * K;
#pragma unroll
      for (int h = 0; h < SW; h++) st[h] = h < K ? src[h] : 0ull;
    } else {
#pragma unroll
      for (int h = 0; h < SW; h++) st[h] = 0ull;
    }
    // match against the distinct states found so far: probe from the hash of the state's words
    uint64_t hv = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int h = 0; h < SW; h++)
      if (h < K) hv = (hv ^ st[h]) * 0xFF51AFD7ED558CCDull + (hv >> 29);
    const unsigned home = (unsigned)(hv >> 32) & hmask;
    int found = -1;
    if (active) {
      unsigned slot = home;
      for (int tries = 0; tries < HS; tries++) {
        const int e = hidx[slot];
        if (e < 0) break;
        bool eq = true;
#pragma unroll
        for (int h = 0; h < SW; h++)
          if (h < K) eq = eq && (uw[(size_t)e * K + h] == st[h]);
        if (eq) {
          found = e;
          break;
        }
        slot = (slot + 1) & hmask;
      }
    }
    // unresolved states become new distinct states in lane (= appearance) order
    unsigned long long pending = __ballot(active && found < 0);
    while (pending) {
      const int leader = __ffsll((long long)pending) - 1;
      bool eq = true;
#pragma unroll
      for (int h = 0; h < SW; h++) {
        if (h < K) {
          const uint64_t lw = __shfl(st[h], leader, WAVE);
          eq = eq && (lw == st[h]);
          if (lane == leader && n_u < CAP) uw[(size_t)n_u * K + h] = lw;
        }
      }
      if (active && found < 0 && eq) found = n_u < CAP ? n_u : CAP;
      if (lane == leader && n_u < CAP) {  // (one lane at a time writes the index: no race; the states beyond CAP are not indexed)
        unsigned slot = home;
        while (hidx[slot] >= 0) slot = (slot + 1) & hmask;
        hidx[slot] = n_u;
      }
      if (n_u < CAP) n_u++;
      else overflow++;
      pending = __ballot(active && found < 0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    if (active && found >= 0 && found < CAP) atomicAdd(&ucount[found], 1);
    __syncthreads();
  }

  // rank by (count desc, first appearance desc)
  for (int e = lane; e < n_u; e += WAVE) {
    const int ce = ucount[e];
    int rank = 0;
    for (int f = 0; f < n_u; f++) {
      const int cf = ucount[f];
      rank += (cf > ce || (cf == ce && f > e)) ? 1 : 0;
    }
    order[rank] = e;
  }
  __syncthreads();

  // support labels in ranked order: label[r] = first rank with the same set of distinct haplotypes
  for (int r = lane; r < n_u; r += WAVE) {
    const uint64_t *a = uw + (size_t)order[r] * K;
    int lab = r;
    for (int q = 0; q < r; q++) {
      const uint64_t *b = uw + (size_t)order[q] * K;
      // sorted haplotypes (W words each): equal supports <=> equal sequences of distinct haplotypes
      int ia = 0, ib = 0;
      bool same = true;
      while (same && (ia < Kh || ib < Kh)) {
        if (ia >= Kh || ib >= Kh) {
          same = false;
          break;
        }
        if (!post_hap_eq(a + ia * W, b + ib * W, W)) {
          same = false;
          break;
        }
        const uint64_t *v = a + ia * W;
        const int ia0 = ia;
        while (ia < Kh && post_hap_eq(a + ia * W, a + ia0 * W, W)) ia++;
        while (ib < Kh && post_hap_eq(b + ib * W, v, W)) ib++;
      }
      if (same) {
        lab = q;
        break;
      }
    }
    label[r] = lab;
  }
  __syncthreads();
  // support sums (sequential in ranked order, as the reference's dict accumulation) and their arg max
  double best = -1.0;
  int best_r = 0x7fffffff;
  const double total = (double)N;
  for (int r = lane; r < n_u; r += WAVE) {
    if (label[r] != r) continue;
    double sum = 0.0;
    bool first = true;
    for (int q = r; q < n_u; q++) {
      if (label[q] != r) continue;
      const double p = (double)ucount[order[q]] / total;
      sum = first ? p : sum + p;
      first = false;
    }
    if (sum > best) {
      best = sum;
      best_r = r;
    }
  }
  // wave arg max, ties -> smallest rank (np.argmax over supports in order of appearance)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ob = __shfl_xor(best, o, WAVE);
    const int orr = __shfl_xor(best_r, o, WAVE);
    if (ob > best || (ob == best && orr < best_r)) {
      best = ob;
      best_r = orr;
    }
  }
  PostSummary R;
  R.n_u = n_u;
  R.overflow = overflow;
  R.best = best;
  R.best_r = best_r;
  return R;
}

template <int SW>
__global__ __launch_bounds__(64) void trace_posterior_kernel(const PosteriorParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int row = blockIdx.x;                                   // row of the per-state outputs
  const int unit = P.unit_list ? P.unit_list[row] : row;        // index of the per-unit outputs
  const mchap_unit U = P.units[unit];
  const int W = P.wph;
  const int K = U.ploidy * W;          // words of a state
  const int KWmax = P.ploidy_max * W;  // ... and of an output row
  const int lane = threadIdx.x;
  const int N = P.chains * (P.steps - P.burn);
  const int CAP = P.cap;
  if (U.ploidy > P.ploidy_max || U.ploidy < 1 || K > SW) {  // the LDS tables are sized for ploidy_max: refuse instead of overrunning them
    if (lane == 0) {
      P.post_n[unit] = INT_MIN;
      P.mode_stats[2 * (size_t)unit + 0] = NAN;
      P.mode_stats[2 * (size_t)unit + 1] = NAN;
      P.mode_index[unit] = -1;
      if (P.mode_count) P.mode_count[unit] = 0;
    }
    return;
  }
  uint64_t *uw = reinterpret_cast<uint64_t *>(smem);                  // [CAP][K]
  int *ucount = reinterpret_cast<int *>(smem + (size_t)CAP * K * 8);  // [CAP]
  int *order = ucount + CAP;                                          // [CAP] rank -> unique index
  int *label = order + CAP;                                           // [CAP] support label by rank
  const int HS = post_hash_slots(CAP);
  int *hidx = label + CAP;                                            // [HS] hash index of the distinct states
  const PostSummary R = post_summarise<SW>(P.trace, U, P.steps, P.burn, 0, P.chains, uw, ucount, order, label, CAP, W, hidx, HS);
  const int n_u = R.n_u;
  for (int r = lane; r < n_u; r += WAVE) {
    if (r < P.max_states) {
      const int e = order[r];
      uint64_t *dst = P.post_words + ((size_t)row * P.max_states + r) * KWmax;
      for (int h = 0; h < KWmax; h++) dst[h] = h < K ? uw[(size_t)e * K + h] : 0ull;
      P.post_counts[(size_t)row * P.max_states + r] = ucount[e];
    }
  }
  for (int r = n_u + lane; r < P.max_states; r += WAVE) {
    uint64_t *dst = P.post_words + ((size_t)row * P.max_states + r) * KWmax;
    for (int h = 0; h < KWmax; h++) dst[h] = 0ull;
    P.post_counts[(size_t)row * P.max_states + r] = 0;
  }
  if (lane == 0) {
    P.post_n[unit] = R.overflow ? -(n_u + R.overflow) : n_u;
    if (n_u > 0) {
      P.mode_stats[2 * (size_t)unit + 0] = R.best;                                            // SPM
      P.mode_stats[2 * (size_t)unit + 1] = (double)ucount[order[R.best_r]] / (double)N;       // GPM
      P.mode_index[unit] = R.best_r;
      if (P.mode_words)
        for (int h = 0; h < KWmax; h++)
          P.mode_words[(size_t)unit * KWmax + h] = h < K ? uw[(size_t)order[R.best_r] * K + h] : 0ull;
      if (P.mode_count) P.mode_count[unit] = ucount[order[R.best_r]];
    } else {
      if (P.mode_count) P.mode_count[unit] = 0;
      P.mode_stats[2 * (size_t)unit + 0] = NAN;
      P.mode_stats[2 * (size_t)unit + 1] = NAN;
      P.mode_index[unit] = -1;
    }
  }
}

// Replicate incongruence of the chains of a unit (GenotypeMultiTrace.replicate_incongruence, assemble/classes.py:341-376,
// the MCI field of `mchap assemble`): per chain the mode support and its probability; among the chains whose mode
// support reaches `threshold`, 0 if they agree on the support's set of haplotypes, else 1, or 2 if together they hold
// more distinct haplotypes than the first of those chains' supports.  -1 if a chain visited more than POST_CAP distinct states.
constexpr int POST_MAX_CHAINS = 32;
struct IncongruenceParams {
  const mchap_unit *units;
  const uint64_t *trace;
  int steps, chains, burn;
  double threshold;
  int ploidy_max;
  int wph;                   // as PosteriorParams::wph
  int calling;               // 1: the traces are `mchap call`'s (a word = an allele index) and the code is the calling classes'
                             // (calling/classes.py:231-263): chains are compared by their mode GENOTYPES, and the code is 2 when the
                             // genotypes together hold more distinct alleles than the ploidy
  int cap;                   // as PosteriorParams::cap
  const int32_t *unit_list;  // null, or the units to summarise (mci is indexed by unit)
  int32_t *mci;
};

template <int SW>
__global__ __launch_bounds__(64) void trace_incongruence_kernel(const IncongruenceParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int unit = P.unit_list ? P.unit_list[blockIdx.x] : (int)blockIdx.x;
  const mchap_unit U = P.units[unit];
  const int W = P.wph;
  const int Kh = U.ploidy, K = Kh * W;  // haplotypes / words of a state
  const int lane = threadIdx.x;
  const int CAP = P.cap;
  if (Kh > P.ploidy_max || Kh < 1 || K > SW) {
    if (lane == 0) P.mci[unit] = -1;
    return;
  }
  uint64_t *uw = reinterpret_cast<uint64_t *>(smem);
  int *ucount = reinterpret_cast<int *>(smem + (size_t)CAP * K * 8);
  int *order = ucount + CAP;
  int *label = order + CAP;
  const int HS = post_hash_slots(CAP);
  int *hidx = label + CAP;
  // [chains][SW] words: the distinct haplotypes of the chain's mode support, W words each
  uint64_t *sets = reinterpret_cast<uint64_t *>(smem + (((size_t)CAP * K * 8 + (size_t)CAP * 12 + (size_t)HS * 4 + 7) & ~(size_t)7));
  int *nset = reinterpret_cast<int *>(sets + (size_t)POST_MAX_CHAINS * SW);  // [chains] size, 0 = below threshold
  int bad = 0;
  for (int ch = 0; ch < P.chains; ch++) {
    const PostSummary R = post_summarise<SW>(P.trace, U, P.steps, P.burn, ch, ch + 1, uw, ucount, order, label, CAP, W, hidx, HS);
    if (R.overflow) bad = 1;
    if (lane == 0) {
      int n = 0;
      if (R.n_u > 0 && R.best >= P.threshold) {
        const uint64_t *g = uw + (size_t)order[R.best_r] * K;  // sorted haplotypes
        for (int h = 0; h < Kh; h++)
          if (P.calling || h == 0 || !post_hap_eq(g + h * W, g + (h - 1) * W, W)) {  // (calling: the genotype itself, copies included)
            for (int w = 0; w < W; w++) sets[(size_t)ch * SW + n * W + w] = g[h * W + w];
            n++;
          }
      }
      nset[ch] = n;
    }
    __syncthreads();
  }
  if (lane == 0) {
    int out = 0;
    int first = -1, modes = 0;
    // number of distinct allele sets among the qualifying chains
    for (int a = 0; a < P.chains; a++) {
      if (nset[a] == 0) continue;
      bool seen = false;
      for (int b = 0; b < a && !seen; b++) {
        if (nset[b] != nset[a]) continue;
        bool eq = true;
        for (int i = 0; i < nset[a] * W; i++) eq = eq && sets[(size_t)a * SW + i] == sets[(size_t)b * SW + i];
        seen = eq;
      }
      if (!seen) modes++;
      if (first < 0) first = a;
    }
    if (modes > 1) {
      out = 1;
      // size of the union of the sets
      int total = 0;
      for (int a = 0; a < P.chains; a++)
        for (int i = 0; i < nset[a]; i++) {
          const uint64_t *w = sets + (size_t)a * SW + i * W;
          bool dup = false;
          for (int b = 0; b < a && !dup; b++)
            for (int q = 0; q < nset[b] && !dup; q++) dup = post_hap_eq(sets + (size_t)b * SW + q * W, w, W);
          for (int q = 0; q < i && !dup; q++) dup = post_hap_eq(sets + (size_t)a * SW + q * W, w, W);  // (copies within a genotype: calling)
          if (!dup) total++;
        }
      // the reference compares with the size of the FIRST qualifying chain's allele set, not with the ploidy
      // (assemble/classes.py:371-375: `ploidy = len(alleles[0])`)
      if (total > nset[first]) out = 2;
    }
    P.mci[unit] = bad ? -1 : out;
  }
}

// (kw: words of a state = ploidy bound x words per haplotype; sw: the kernels' SW)
inline size_t posterior_lds_bytes(int kw, int cap = POST_CAP) {
  return (((size_t)cap * kw * 8 + (size_t)cap * 4 * 3 + (size_t)post_hash_slots(cap) * 4) + 7) & ~(size_t)7;
}
inline size_t incongruence_lds_bytes(int kw, int cap = POST_CAP, int sw = MCHAP_MAX_PLOIDY) {
  return posterior_lds_bytes(kw, cap) + (size_t)POST_MAX_CHAINS * sw * 8 + (size_t)POST_MAX_CHAINS * 4;
}
// the largest table (distinct states) a workgroup's 160 KB of LDS holds at this state width, beside the incongruence kernel's sets
inline int posterior_max_cap(int kw, int sw = MCHAP_MAX_PLOIDY) {
  const size_t fixed = (size_t)POST_MAX_CHAINS * sw * 8 + (size_t)POST_MAX_CHAINS * 4 + 64;
  int cap = (int)((160 * 1024 - fixed) / ((size_t)kw * 8 + 12 + 8));  // (+ the hash index: two slots a state, up to four ...)
  while (cap > 1 && posterior_lds_bytes(kw, cap) + fixed > 160 * 1024) cap--;  // ... when 2 cap is just above a power of two
  return cap;
}
constexpr int POST_SW_WIDE = 32;  // ploidies up to 15 at two words per haplotype

}  // namespace mchap
