// De-novo MCMC sampler, "SIMT chains" form for MI355X (gfx950): one LANE per chain, 64 chains per wavefront.
//
// Why.  On the benchmark shape (and on real pileups) a converged chain keeps proposing the same neighbour
// genotypes: with the reference's own likelihood cache (assemble/likelihood.py:151-305) 99.6 % of the requested
// likelihood evaluations are cache hits (69.2 k requests -> 280 evaluations per locus, measured with the CPU
// port).  What remains is the sequential book-keeping of ~40 dependent sub-steps per MCMC step: integer label /
// dosage logic, one exp, one uniform draw, one cache probe.  A wavefront per chain (denovo_kernel.hpp) executes
// that scalar work on 1 of 64 lanes; here every lane runs its own chain, so the same instruction stream advances
// 64 chains, and the wavefront only co-operates (lanes over reads, wave shuffle reduction) when a lane misses its
// cache.
//
// Pipeline: denovo_prepare_kernel (one workgroup per unit: homozygous fix, read tensor transposed to
// [M0*A][RPAD] with NaN -> 1 in the workspace, initial-genotype distribution, prior tables) ->
// denovo_simt_kernel (one wavefront per 64 chains).  Same algorithm, same Philox streams and therefore the same
// traces as denovo_mcmc_kernel; reference citations as in denovo_kernel.hpp.
#pragma once
#include "denovo_kernel.hpp"

namespace mchap {

// per-unit metadata written by the prepare kernel
constexpr int META_I_MH = 0;      // number of sampled (non-fixed) positions
constexpr int META_I_STATUS = 1;  // MCHAP_UNIT_*
constexpr int META_I_NDICT = 2;   // distinct values of the unit's table (0: more than DICT_MAX, no coded table)
constexpr int META_I_FLAT = 3;    // 1: every entry of the unit's table (existing alleles) is a gap -- a sample without reads
                                  // at the locus, which the reference samples all the same (assemble/mcmc.py:132-137): the
                                  // likelihood of every genotype is then the same number
constexpr int META_I_W01 = 4;     // 1: every read weight of the unit is 0 or 1 (no counts of de-duplicated rows): the lanes take ONE
                                  // logarithm per group of up to four reads (read_log_sum, read_log.hpp); tuning flag 1048576: never
constexpr int META_I_COLS = 5;    // then [M]: column (j * A) of sampled position jj ; then [M]: n_alleles
__host__ __device__ inline int meta_i_stride(int max_pos) { return 5 + 2 * max_pos; }
// SpecLds::ndict and its like hold the dictionary size with the unit's META_I_W01 in the top bit
constexpr uint16_t ND_W01 = 0x8000u;
__host__ __device__ inline int nd_count(uint16_t x) { return (int)(x & 0x7FFFu); }
__host__ __device__ inline bool nd_w01(uint16_t x) { return (x & ND_W01) != 0; }
// Coded read table (speculative sampler): a unit's table usually holds a few dozen distinct probabilities
// (one per base quality, its error share, 1.0 for gaps), so it is also stored as uint8 codes into a per-unit
// dictionary of float64 values: lossless, 8x smaller, and what the likelihood evaluation then streams stays in L2.
constexpr int DICT_MAX = 128;  // entries kept per unit (LDS of the sampler: 1 KB per chain)
constexpr int DICT_HASH = 512;   // open-addressing slots of the prepare pass's LDS set (at most DICT_MAX = 128 are taken)
// doubles: [0] luh, [1..] prior table (2K+5), then dist [M*A]
__host__ __device__ inline int meta_f_prior(int) { return 1; }
__host__ __device__ inline int meta_f_dist(int max_ploidy) { return 1 + 2 * max_ploidy + 5; }
__host__ __device__ inline int meta_f_stride(int max_ploidy, int max_pos, int max_allele) {
  return 1 + 2 * max_ploidy + 5 + max_pos * max_allele;
}

struct SimtParams {
  DenovoParams d;
  // workspace carved by the host
  double *rt;        // [U][max_ma][rpad]
  double *cntw;      // [U][rpad]
  uint8_t *codes;    // [U][max_ma][64][cstride]: code of read lane + 64 i at byte i of the lane's group
  double *dict;      // [U][DICT_MAX]
  double *gbp;       // [U * chains][max_ploidy][rpad] haplotype products of the chains' current genotypes for read chunks >= 4, or null
  int32_t *meta_i;   // [U][meta_i_stride]
  double *meta_f;    // [U][meta_f_stride]
  int n_units;
  int max_pos, max_allele, max_ploidy;
  int max_ma;        // max over units of n_pos * max_allele
  int max_ugens_pad; // doubles reserved for the SNV posterior scratch of the prepare pass
  int prep_rows_off; // byte offset of the prepare pass's per-position row staging in its LDS
  int cstride;       // bytes per lane and row of the coded table: rpad / 64 rounded up to 1, 2 or a multiple of 4
  int flags;         // debugging: bit 0 no mutation memo, bit 1 no interval memo (MCHAP_HIP_FLAGS); bit 30 below
  // steady-state pipeline (denovo_lane_kernel.hpp): per-chain hand-over records and threshold tables
  void *lane_state;  // [U * chains] LaneState
  void *lane_memo;   // [U * chains][2][tri(max_pos)] uint32
  // phased sampler (kernel 5: denovo_spec_kernel<K, G, true> in phases + denovo_coast_kernel): hand-over records
  void *pipe_state;           // [U * chains] PipeState
  double *pipe_memo;          // [U * chains][2][tri(max_pos)] total move probabilities of the interval steps
  const int32_t *pipe_list;   // chains of this launch (nullptr: all of them, in order)
  const int32_t *pipe_count;  // their number, on the device (nullptr: all)
  int32_t *pipe_out;          // coasting kernel: the chains it hands back, appended in any order
  int32_t *pipe_out_count;
  int pipe_iters;             // compound steps per chain in this launch (<= 0: to the end)
  int pipe_iters_max;         // ... which a wave extends to while one of its chains is not settled (PIPE_EXPORT)
  int pipe_mode;              // PIPE_RESUME | PIPE_EXPORT | PIPE_FILLONLY
  int pipe_parts;             // wavefronts a short list of chains may spread the completion of one chain's tables over
  int bp_cache;               // speculative sampler with one chain per wave: base-product cache carved after its LDS
  int fill_lt, fill_kw;       // denovo_fill_kernel: lanes per tile of the read table, words kept per distinct request
  int tw_lds;                 // denovo_spec_kernel<.., TW>: bytes of one wavefront's LDS layout behind the workgroup's exchange area
  int word_bits;              // bits of a packed haplotype word of this launch's sampler: 0 / 64, or 128 (denovo_simt_kernel<0, u128>)
  uint64_t cache_epoch;       // speculative / phased sampler: this call's epoch << 33, OR-ed into every likelihood-cache tag (0: the
                              // caches were cleared for this call instead): entries of earlier calls never match, nothing is cleared
  uint64_t *ctx;              // phased sampler, resumed chains: [U * chains][ctx_n][spec_ctx_words] decision contexts per genotype
  int ctx_n;                  // (denovo_spec_kernel.hpp "decision contexts"), or null / 0
};
constexpr int PIPE_RESUME = 1;  // start from the chains' PipeState records
constexpr int PIPE_EXPORT = 2;  // at the end: complete the interval memo of the current genotype, write the records
// Completing a chain's tables is a few hundred likelihood evaluations served one after the other by its wavefront.
// When a launch holds few chains (a list of handed-back chains; a small batch of a big shape) the chip is empty
// and that latency is all there is: the exporting launch then leaves the tables as they are, and a PIPE_FILLONLY
// launch follows in which pipe_parts_eff() wavefronts per chain each complete every pipe_parts_eff()-th unknown entry
// (no steps, no records; the chain's likelihood cache is not used: the wavefronts would race on it).
constexpr int PIPE_FILLONLY = 4;
constexpr int PIPE_NOFILL = 8;  // PIPE_EXPORT without the table completion: denovo_fill_kernel (one lane per request) follows
constexpr int PIPE_FILL_SLOTS = 1024;  // wavefront slots such a launch may occupy (one per SIMD: beyond that the chip is busy anyway)
__host__ __device__ inline int pipe_parts_eff(int n_list, int parts) {
  if (parts <= 1 || n_list <= 0) return 1;
  const int pe = PIPE_FILL_SLOTS / n_list;
  return pe < 1 ? 1 : (pe > parts ? parts : pe);
}
// hand-over record of a chain between the launches of the phased sampler
struct PipeState {
  uint64_t g[8];     // haplotype words in the chain's own (unsorted) order
  double llk;
  uint64_t ctr;      // next draw of the chain's stream
  double mlo, mhi;   // mutation step: no move while every uniform is in [mlo, mhi)
  int32_t step;      // MCMC steps done (== steps: finished, or stopped by an error status)
  int32_t mvalid;
  uint64_t pad[3];
};
static_assert(sizeof(PipeState) == 128, "PipeState layout");
constexpr int SIMT_FLAG_PREP_GLOBAL = 1 << 30;  // table too large for the prepare pass's LDS copy

// ---------------------------------------------------------------------------------------------------------
// prepare: grid = units, block = 64
// ---------------------------------------------------------------------------------------------------------
#ifndef MCHAP_PREP_WPE
#define MCHAP_PREP_WPE 4  // waves per SIMD the prepare pass is compiled for (it is latency-bound: one wave per unit)
#endif
template <int RPL>
__global__ __launch_bounds__(64, MCHAP_PREP_WPE) void denovo_prepare_kernel(const SimtParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int u = blockIdx.x;
  const mchap_unit U = D.units[u];
  const int lane = threadIdx.x;
  const int R = U.n_reads, M0 = U.n_pos, A = U.max_allele, K = U.ploidy;
  const int rpad = D.rpad;
  const int MA = M0 * A;
  double *rt = P.rt + (size_t)u * P.max_ma * rpad;
  // [MA][rpad] copy for the homozygous fix: in LDS when it fits, else the global table itself (every lane only
  // re-reads the reads r = lane + 64 i it stored, so program order makes its own stores visible)
  const bool in_lds = !(P.flags & SIMT_FLAG_PREP_GLOBAL);
  double *rl = in_lds ? reinterpret_cast<double *>(smem) : rt;
  double *lp = reinterpret_cast<double *>(smem) + (in_lds ? (size_t)MA * rpad : 0);  // snv posterior scratch
  const double *gr = D.reads ? D.reads + U.reads_off : nullptr;
  const int8_t *nal0 = D.n_alleles + U.nalleles_off;
  // entry (r, q = j * A + a) of the unit's probability tensor: read from it, or formed from the allele calls as
  // encoding/integer/transcode.py:16-77 does (called allele p, the others (1 - p) / 3, NaN for a gap, then 0 for
  // alleles the position does not have)
  auto rawv = [&](int r, int q) -> double {
    if (gr) return gr[(size_t)r * MA + q];
    const int j = q / A, a = q - j * A;
    if (a >= nal0[j]) return 0.0;
    const size_t e = (size_t)U.reads_off + (size_t)r * M0 + j;
    const int call = D.calls[e];
    if (call < 0) return NAN;
    int qi = D.quals ? (int)D.quals[e] : 0;
    qi = qi < 0 ? 0 : (qi >= D.qual_prob_len ? D.qual_prob_len - 1 : qi);
    const double pc = D.qual_prob[qi];
    return a == call ? pc : (1.0 - pc) / 3.0;
  };
  double *cw = P.cntw + (size_t)u * rpad;
  int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);

  bool all_gaps = true;  // every factor a likelihood can read is 1.0
  for (int r = lane; r < rpad; r += WAVE) {
    // eight entries of the read's row at a time: the loads are independent and issued together (the stores that
    // follow could alias them as far as the compiler knows)
    for (int q0 = 0; q0 < MA; q0 += 8) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (r < R && q0 + k < MA) ? rawv(r, q0 + k) : 1.0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int q = q0 + k;
        if (q < MA) {
          const double x = isnan(v[k]) ? 1.0 : v[k];
          if (in_lds) rl[(size_t)q * rpad + r] = x;
          rt[(size_t)q * rpad + r] = x;
          if (q % A < (int)nal0[q / A]) all_gaps = all_gaps && (x == 1.0);
        }
      }
    }
  }
  {
    const bool flat = __ballot(!all_gaps) == 0ull;
    if (lane == 0) mi[META_I_FLAT] = flat ? 1 : 0;
  }
  double cnt[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) {
    const int r = lane + WAVE * i;
    cnt[i] = (r < R) ? (U.counts_off >= 0 ? (double)D.counts[U.counts_off + r] : 1.0) : 0.0;
    cw[r] = cnt[i];
  }
  {
    bool w01 = true;
#pragma unroll
    for (int i = 0; i < RPL; i++) w01 = w01 && (cnt[i] == 0.0 || cnt[i] == 1.0);
    const bool all01 = __ballot(!w01) == 0ull;
    if (lane == 0) mi[META_I_W01] = (all01 && !(P.flags & (1 << 20))) ? 1 : 0;
  }
  __syncthreads();
  // ---- dictionary + coded table ----
  {
    unsigned long long *hkeys = reinterpret_cast<unsigned long long *>(lp + P.max_ugens_pad);  // [DICT_HASH]
    uint16_t *hcode = reinterpret_cast<uint16_t *>(hkeys + DICT_HASH);                         // [DICT_HASH]
    int *ndist = reinterpret_cast<int *>(hcode + DICT_HASH);
    const unsigned long long EMPTY = ~0ull;  // a NaN pattern: table entries are never NaN
    for (int i = lane; i < DICT_HASH; i += WAVE) hkeys[i] = EMPTY;
    if (lane == 0) *ndist = 0;
    __syncthreads();
    const int RPLT = rpad / WAVE;
    for (int q = 0; q < MA; q++) {
      // the lane's RPL values of this row: loaded together (one memory round trip per row, not one per value)
      for (int i0 = 0; i0 < RPL; i0 += 4) {
      unsigned long long keys[4];
#pragma unroll
      for (int k = 0; k < 4; k++) keys[k] = (unsigned long long)__double_as_longlong(rt[(size_t)q * rpad + lane + WAVE * min(i0 + k, RPL - 1)]);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (i0 + k >= RPL) break;
        const unsigned long long key = keys[k];
        unsigned slot = (unsigned)(mix64(key) & (DICT_HASH - 1));
        while (*(volatile int *)ndist <= DICT_MAX) {
          // plain read first: most entries repeat a value that is already in the set (same-address LDS atomics of a
          // whole wavefront serialise)
          unsigned long long old = *(volatile unsigned long long *)&hkeys[slot];
          if (old == EMPTY) {
            old = atomicCAS(&hkeys[slot], EMPTY, key);
            if (old == EMPTY) atomicAdd(ndist, 1);
          }
          if (old == EMPTY || old == key) break;
          slot = (slot + 1) & (DICT_HASH - 1);
        }
      }
      }
    }
    __syncthreads();
    const int nd = *ndist;
    uint8_t *ct = P.codes + (size_t)u * P.max_ma * WAVE * P.cstride;
    double *dict = P.dict + (size_t)u * DICT_MAX;
    if (nd <= DICT_MAX) {
      int base = 0;
      for (int s0 = 0; s0 < DICT_HASH; s0 += WAVE) {
        const unsigned long long key = hkeys[s0 + lane];
        const bool occ = key != EMPTY;
        const unsigned long long m = __ballot(occ);
        const int code = base + __popcll(m & ((1ull << lane) - 1ull));
        if (occ) {
          hcode[s0 + lane] = (uint16_t)code;
          dict[code] = __longlong_as_double((long long)key);
        }
        base += __popcll(m);
      }
      __syncthreads();
      for (int q = 0; q < MA; q++) {
        for (int i0 = 0; i0 < RPL; i0 += 4) {
          unsigned long long keys[4];
#pragma unroll
          for (int k = 0; k < 4; k++) keys[k] = (unsigned long long)__double_as_longlong(rt[(size_t)q * rpad + lane + WAVE * min(i0 + k, RPL - 1)]);
          uint32_t packed = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (i0 + k >= RPL) break;
            const unsigned long long key = keys[k];
            unsigned slot = (unsigned)(mix64(key) & (DICT_HASH - 1));
            while (hkeys[slot] != key) slot = (slot + 1) & (DICT_HASH - 1);
            if (RPL > 2) packed |= (uint32_t)(uint8_t)hcode[slot] << (8 * k);
            else ct[((size_t)q * WAVE + lane) * P.cstride + i0 + k] = (uint8_t)hcode[slot];
          }
          // (cstride is a multiple of 4 beyond two chunks: one aligned 4-byte store instead of four byte stores;
          // bytes past the unit's chunks are padding the sampler never reads)
          if (RPL > 2) *reinterpret_cast<uint32_t *>(ct + ((size_t)q * WAVE + lane) * P.cstride + i0) = packed;
        }
      }
    }
    if (lane == 0) mi[META_I_NDICT] = nd <= DICT_MAX ? nd : 0;
  }
  const int8_t *nalleles = D.n_alleles + U.nalleles_off;
  // homozygous fix (assemble/mcmc.py:168-182, 494-541; snpcalling.py:14-70)
  int Mh = 0;
  double luh = 0.0;
  const bool pow2_ploidy = (K & (K - 1)) == 0;
  const double inv_ploidy = 1.0 / (double)K;
  // The SNV prior (snv_log_prior: calling/prior.py:116-179 with flat frequencies) needs lgamma(dose + 1) and, with
  // inbreeding, lgamma(dose + alpha_n), lgamma(alpha_n) and the normaliser for every allele count n: a few dozen
  // distinct values per unit.  Every lane used to recompute K + 1 of them per genotype (200 calls of ~400
  // instructions per unit: 60 % of this pass); now each value is formed once, one per lane.
  constexpr int KP = MCHAP_MAX_PLOIDY_DENOVO;  // (the prepare pass serves every sampler: ploidies up to 15, packs of sixteen nibbles)
  __shared__ double s_lg1[KP + 1];                              // lgamma(d + 1)
  __shared__ double s_lga[(MCHAP_MAX_ALLELE + 1) * (KP + 1)];   // lgamma(d + alpha_n), d >= 1
  __shared__ double s_lgb[MCHAP_MAX_ALLELE + 1], s_left[MCHAP_MAX_ALLELE + 1];  // lgamma(alpha_n); normaliser
  const double Fp = U.inbreeding;
  const bool with_prior = !isnan(Fp);
  if (with_prior) {
    for (int d = lane; d <= K; d += WAVE) s_lg1[d] = lgamma((double)d + 1.0);
    if (Fp != 0.0) {
      for (int e = lane; e < (A + 1) * (K + 1); e += WAVE) {
        const int n = e / (K + 1), d = e % (K + 1);
        if (n >= 1 && d >= 1) s_lga[n * (KP + 1) + d] = lgamma((double)d + (1.0 / (double)n) * ((1.0 - Fp) / Fp));
      }
      for (int n = 1 + lane; n <= A; n += WAVE) {
        const double alpha = (1.0 / (double)n) * ((1.0 - Fp) / Fp);
        const double sum_alphas = alpha * (double)n;
        s_lgb[n] = lgamma(alpha);
        s_left[n] = (lgamma((double)K + 1.0) + lgamma(sum_alphas)) - lgamma((double)K + sum_alphas);
      }
    }
    __syncthreads();
  }
  // snv_log_prior(g, K, n, F) from the tables: same arguments to the same lgamma, same order of the sums
  auto snv_prior = [&](uint64_t g, int n) -> double {
    int dose[KP];
    for (int i = 0; i < K; i++) dose[i] = 0;
    for (int i = 0; i < K; i++) {
      int j = 0;
      while (nib(g, i) != nib(g, j)) j++;
      dose[j] += 1;
    }
    if (Fp == 0.0) {
      double den = 0.0;
      for (int i = 0; i < K; i++) den += s_lg1[dose[i]];
      return (s_lg1[K] - den) - (double)K * c_ln[n];
    }
    double prod = 0.0;
    for (int i = 0; i < K; i++)
      if (dose[i] > 0) prod += s_lga[n * (KP + 1) + dose[i]] - (s_lg1[dose[i]] + s_lgb[n]);
    return s_left[n] + prod;
  };
  // without the LDS copy of the whole table, the A rows of the current position are staged in LDS (each lane its own
  // reads): the genotype loop below re-reads them K times per genotype
  double *pl = reinterpret_cast<double *>(smem + P.prep_rows_off);  // [A][rpad]
  for (int j = 0; j < M0; j++) {
    const int n = nalleles[j];
    const int u_gens = snv_genotypes(n, K);
    const double *rowp = rl + (size_t)(j * A) * rpad;
    if (!in_lds) {
      for (int a = 0; a < A; a++)
#pragma unroll
        for (int i = 0; i < RPL; i++) pl[(size_t)a * rpad + lane + WAVE * i] = rt[(size_t)(j * A + a) * rpad + lane + WAVE * i];
      rowp = pl;
    }
    uint64_t g = 0;  // the SNV genotype's alleles, one nibble each
    for (int q = 0; q < u_gens; q++) {
      double lprior = 0.0;
      if (with_prior) lprior = snv_prior(g, n);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < RPL; i++) {
        double rp = 0.0;
        // x / K (snpcalling.py / likelihood.py:61); for K = 2, 4, 8 the product with 1 / K is the same double
        if (pow2_ploidy) {
          for (int h = 0; h < K; h++) rp += rowp[(size_t)nib(g, h) * rpad + lane + WAVE * i] * inv_ploidy;
        } else {
          for (int h = 0; h < K; h++) rp += rowp[(size_t)nib(g, h) * rpad + lane + WAVE * i] / (double)K;
        }
        s += read_log(rp) * cnt[i];
      }
      const double llk = wave_sum(s);
      lp[q] = lprior + llk;
      g = increment_snv_genotype(g, K);
    }
    double acc = lp[0];
    for (int q = 1; q < u_gens; q++) acc = add_log_prob(acc, lp[q]);
    int fixed_allele = -1;
    for (int a = 0; a < n; a++) {
      int idx = 0;
      for (int i = 0; i < K; i++) idx += (a == 0) ? 0 : snv_genotypes(a, i + 1);
      if (exp(lp[idx] - acc) >= D.fix_hom) fixed_allele = a;
    }
    if (lane == 0) D.fixed[U.fixed_off + j] = (int8_t)fixed_allele;
    if (fixed_allele < 0) {
      if (lane == 0) {
        mi[META_I_COLS + Mh] = j * A;
        mi[META_I_COLS + P.max_pos + Mh] = n;
      }
      luh += c_ln[n];  // assemble/mcmc.py:294
      Mh++;
    }
  }
  const int bits = allele_bits(A);
  int status = MCHAP_UNIT_OK;
  if (Mh == 0) status = MCHAP_UNIT_ALL_FIXED;
  else if (Mh * bits > (P.word_bits ? P.word_bits : 64)) status = MCHAP_ERR_LIMIT;
  else if (U.initial_off >= 0 && U.initial_n_het != Mh) status = MCHAP_UNIT_BAD_INITIAL;  // assemble/mcmc.py:207
  if (lane == 0) {
    mi[META_I_MH] = Mh;
    mi[META_I_STATUS] = status;
    D.status[u] = status;
    mf[0] = luh;
  }
  if (status != MCHAP_UNIT_OK) {
    if (status == MCHAP_UNIT_ALL_FIXED) {
      // assemble/mcmc.py:189-199: constant trace, NaN llks
      const size_t tb = U.trace_off, lb = U.llk_off;
      const int S = D.steps, C_ = D.chains;
      for (int i = lane; i < C_ * S * K; i += WAVE) D.trace[tb + i] = 0ull;
      for (int i = lane; i < C_ * S; i += WAVE) D.llks[lb + i] = NAN;
    }
    return;
  }
  // prior tables (assemble/prior.py:39-112), same layout as Chain::prior_tab
  if (!isnan(U.inbreeding) && lane == 0) {
    double *t = mf + meta_f_prior(0);
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
  // _read_mean_dist (assemble/mcmc.py:455-491) over the sampled positions, from the raw tensor
  if (U.initial_off < 0) {
    double *dist = mf + meta_f_dist(P.max_ploidy);
    __syncthreads();
    for (int jj = 0; jj < Mh; jj++) {
      const int col = mi[META_I_COLS + jj];
      int n_nonzero = 0;
      uint32_t gapmask = 0;
      double dv[MCHAP_MAX_ALLELE];
      for (int a = 0; a < A; a++) {
        double tot = 0.0;
        int n_ok = 0, n_nz = 0;
        for (int i0 = 0; i0 < RPL; i0 += 4) {  // four of the lane's reads at a time: their loads are independent
          double v4[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int r = lane + WAVE * (i0 + k);
            v4[k] = (i0 + k < RPL && r < R) ? rawv(r, col + a) : 0.0;
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int r = lane + WAVE * (i0 + k);
            if (i0 + k < RPL && r < R) {
              const double v = v4[k];
              if (!isnan(v)) {
                tot += v;
                n_ok++;
              }
              if (!(v == 0.0)) n_nz++;
            }
          }
        }
        tot = wave_sum(tot);
        n_ok = wave_sum_i(n_ok);
        n_nz = wave_sum_i(n_nz);
        if (n_ok == 0) {
          gapmask |= 1u << a;
          dv[a] = 1.0;
          n_nonzero++;
        } else {
          dv[a] = tot / (double)n_ok;
          if (n_nz > 0) n_nonzero++;
        }
      }
      for (int a = 0; a < A; a++)
        if (gapmask & (1u << a)) dv[a] = 1.0 / (double)n_nonzero;
      double s = 0.0;
      for (int a = 1; a < A; a++) s += dv[a];
      s = (A > 1) ? dv[0] + s : dv[0];
      if (lane == 0)
        for (int a = 0; a < A; a++) dist[jj * A + a] = dv[a] / s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// sampler: one lane per chain
// ---------------------------------------------------------------------------------------------------------
// W: the type of a packed haplotype word -- uint64_t (every shape of rounds 1-3: at most 64 bits of sampled alleles per haplotype),
// or unsigned __int128 (round 4: up to 128 bits, i.e. 126 biallelic / 64 tri- or tetra-allelic SNVs; `denovo_simt_kernel<0, u128>`,
// the general fallback for targets wider than the fast samplers take).  The code is the same; only the word's type differs.
typedef unsigned __int128 u128;
template <class W> struct WordBits { static constexpr int value = 8 * (int)sizeof(W); };
__device__ __forceinline__ int word_popc(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int word_popc(u128 x) { return __popcll((uint64_t)x) + __popcll((uint64_t)(x >> 64)); }
__device__ __forceinline__ int word_ffs(uint64_t x) { return __ffsll((long long)x); }  // 1-based, 0 if none
__device__ __forceinline__ int word_ffs(u128 x) {
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  return lo ? __ffsll((long long)lo) : (hi ? 64 + __ffsll((long long)hi) : 0);
}
__device__ __forceinline__ uint64_t word_mix(uint64_t t, uint64_t x) { return mix64(t ^ x) + 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ uint64_t word_mix(uint64_t t, u128 x) {
  t = mix64(t ^ (uint64_t)(x >> 64)) + 0x9E3779B97F4A7C15ull;
  return mix64(t ^ (uint64_t)x) + 0x9E3779B97F4A7C15ull;
}

// nibble packs of per-haplotype labels / doses: eight nibbles (ploidy <= 8) beside 64-bit words, sixteen (ploidy <= 15: a dose of
// 16 would not fit a nibble) beside 128-bit words -- the general instantiation takes both the wide and the high-ploidy units
template <class W> struct PackOf { typedef uint32_t type; static constexpr int KMAX = MCHAP_MAX_PLOIDY; };
template <> struct PackOf<u128> { typedef uint64_t type; static constexpr int KMAX = MCHAP_MAX_PLOIDY_DENOVO; };

template <class W>
struct SimtLdsT {
  W *w;             // [T*Kmax][64]
  W *pw;            // [Kmax][64]
  double *llk_t;    // [T][64]
  uint64_t *rngn;   // [T][64]
  double *prior;    // [2Kmax+5][64]
  uint8_t *optin;   // [Kmax*Kmax][64] option i of the lane's interval step as (h0 << 4) | h1: the move, not the label pack it yields
  uint16_t *sub;    // [Kmax*Mmax][64]
  uint16_t *cols;   // [Mmax][64]
  uint8_t *shift;   // [Mmax][64]
  uint8_t *nal;     // [Mmax][64]
};
typedef SimtLdsT<uint64_t> SimtLds;

__host__ __device__ inline size_t simt_lds_bytes(int Kmax, int Mmax, int T, int word_bytes = 8) {
  size_t b = 0;
  b += (size_t)word_bytes * T * Kmax * 64;
  b += (size_t)word_bytes * Kmax * 64;
  b += (size_t)8 * T * 64 * 2;
  b += (size_t)8 * (2 * Kmax + 5) * 64;
  b += (size_t)Kmax * Kmax * 64;
  b += (size_t)2 * Kmax * Mmax * 64;
  b += (size_t)2 * Mmax * 64;
  b += (size_t)1 * Mmax * 64 * 2;
  return (b + 15) & ~(size_t)15;
}

struct Lane {
  int K, Mh, bits, key_bits;
  uint32_t amask;
  double invK, inbreeding;
  bool alive;            // lane owns a chain that is still running
  const double *rt;      // unit's transposed reads [ma][rpad]
  const double *cw;      // unit's counts [rpad]; bit 0 of the pointer: the unit's META_I_W01 (its weights are 0 / 1: one logarithm per
                         // group of four chunks) -- a field of its own here made the 128-bit instantiation fault (round 5)
  ulonglong2 *cache;
  uint64_t *ckeys;  // words of the cached genotypes when they are wider than the tag (else nullptr)
  int key_words;
  uint32_t cache_mask;
  Rng rng;
};

#define L_(arr, i) (arr)[(size_t)(i) * WAVE + lane]

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
  return v;
}

// per-lane helpers on the lane's haplotype words (row `base` .. base+K-1 of an LDS [..][64] array)
// KT > 0: compile-time ploidy shared by every chain of the launch (loops unroll to exactly K); KT == 0: per-lane K
template <int KT, class W>
__device__ __forceinline__ typename PackOf<W>::type lane_dosage_of_words(const W *arr, int base, int K, int lane) {
  typedef typename PackOf<W>::type P;
  constexpr int KM = KT ? KT : PackOf<W>::KMAX;
  W x[KM];
#pragma unroll
  for (int h = 0; h < KM; h++) x[h] = h < K ? L_(arr, base + h) : (W)0;
  P d = 0;
#pragma unroll
  for (int h = 0; h < KM; h++)
    if (h < K) d |= (P)1 << (4 * h);
#pragma unroll
  for (int h = 0; h < KM; h++) {
    if (h >= K || nib(d, h) == 0) continue;
#pragma unroll
    for (int p = 0; p < KM; p++) {
      if (p <= h || p >= K || nib(d, p) == 0) continue;
      if (x[h] == x[p]) {
        d += (P)1 << (4 * h);
        d &= ~((P)15 << (4 * p));
      }
    }
  }
  return d;
}

template <int KT, class W>
__device__ __forceinline__ int lane_count_copies(const W *arr, int base, int K, int h, int lane) {
  constexpr int KM = KT ? KT : PackOf<W>::KMAX;
  const W x = L_(arr, base + h);
  int n = 0;
#pragma unroll
  for (int i = 0; i < KM; i++)
    if (i < K) n += (L_(arr, base + i) == x) ? 1 : 0;
  return n;
}

template <class W>
__device__ inline double lane_prior_of_dosage(const SimtLdsT<W> &S, const Lane &c, typename PackOf<W>::type d, int lane) {
  const int K = c.K;
  if (c.inbreeding == 0.0) {
    double den = 0.0;
    for (int i = 0; i < K; i++) den += L_(S.prior, K + 1 + nib(d, i));
    return (L_(S.prior, 2 * K + 3) - den) - L_(S.prior, 2 * K + 4);
  }
  double prod = 0.0;
  for (int i = 0; i < K; i++) {
    const uint32_t dose = nib(d, i);
    if (dose > 0) prod += L_(S.prior, dose);
  }
  return L_(S.prior, 2 * K + 2) + prod;
}

template <int KT, class W>
__device__ __forceinline__ double lane_words_prior(const SimtLdsT<W> &S, const Lane &c, const W *arr, int base, int lane) {
  if (isnan(c.inbreeding)) return 0.0;
  return lane_prior_of_dosage(S, c, lane_dosage_of_words<KT, W>(arr, base, KT ? KT : c.K, lane), lane);
}

template <int KT, class W>
__device__ __forceinline__ typename PackOf<W>::type lane_segment_labels(const W *arr, int base, int K, W mask, int lane) {
  typedef typename PackOf<W>::type P;
  constexpr int KM = KT ? KT : PackOf<W>::KMAX;
  W x[KM];
#pragma unroll
  for (int h = 0; h < KM; h++) x[h] = h < K ? (L_(arr, base + h) & mask) : (W)0;
  P lab = 0;
#pragma unroll
  for (int h = 1; h < KM; h++) {
    if (h >= K) continue;
    int l = h;
#pragma unroll
    for (int g = KM - 1; g >= 0; g--)
      if (g < h && x[g] == x[h]) l = g;  // smallest matching index wins
    lab |= (P)l << (4 * h);
  }
  return lab;
}

template <class W>
__device__ __forceinline__ W lane_interval_mask(const Lane &c, int start, int stop) {
  constexpr int WB = WordBits<W>::value;
  const int nb = c.bits * (stop - start);
  const W ones = nb >= WB ? ~(W)0 : (((W)1 << nb) - (W)1);
  const int sh = c.bits * (c.Mh - stop);
  return sh >= WB ? (W)0 : ones << sh;
}

// Co-operative likelihood evaluation of the proposals S.pw[.][src] of every lane `src` with need set:
// lanes over reads, coalesced 512-byte row reads of the unit's transposed tensor, wave butterfly sum.
// The K*Mh row indices of a request are first computed one per lane and then broadcast with readlane, so the
// global loads depend on nothing but registers and are issued back to back (one exposed L2 latency per request
// instead of one per position).
// one request: RPL chunks of 64 reads per lane; the pair loop is unrolled so that UNR * RPL row loads are in
// flight before the first multiply (one exposed memory latency per UNR pairs, not per pair)
template <int RPL, class W>
__device__ __forceinline__ double coop_body(const SimtLdsT<W> &S, int src, int K, int Mh, uint32_t amask, double invK,
                                            const double *rt, const double *cw, int rpad, int lane, bool grouped) {
  constexpr int UNR = RPL <= 4 ? 8 : (RPL == 8 ? 4 : 2);
  const int n_pairs = K * Mh;
  double acc[RPL], prod[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) {
    acc[i] = 0.0;
    prod[i] = 1.0;
  }
  for (int base = 0; base < n_pairs; base += WAVE) {
    int myrow = 0;  // lane l owns pair base + l = (h, j)
    {
      const int p = base + lane;
      if (p < n_pairs) {
        const int h = p / Mh, j = p - h * Mh;
        const W wh = S.pw[(size_t)h * WAVE + src];
        const uint32_t a = (uint32_t)(wh >> S.shift[(size_t)j * WAVE + src]) & amask;
        myrow = (int)S.cols[(size_t)j * WAVE + src] + (int)a;
      }
    }
    const int lim = min(WAVE, n_pairs - base);
    int jj = base % Mh;  // position of pair `base` inside its haplotype
    for (int q0 = 0; q0 < lim; q0 += UNR) {
      double v[UNR][RPL];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int q = q0 + u;
        const int row = __builtin_amdgcn_readlane(myrow, q < lim ? q : 0);
        const double *rp = rt + (size_t)row * rpad;
#pragma unroll
        for (int i = 0; i < RPL; i++) v[u][i] = rp[WAVE * i];
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        if (q0 + u < lim) {
#pragma unroll
          for (int i = 0; i < RPL; i++) prod[i] *= v[u][i];
          if (++jj == Mh) {  // haplotype complete
            jj = 0;
#pragma unroll
            for (int i = 0; i < RPL; i++) {
              acc[i] += prod[i] * invK;
              prod[i] = 1.0;
            }
          }
        }
      }
    }
  }
  double wv[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) wv[i] = cw[WAVE * i];
  return wave_sum(read_log_sum_chunks<RPL>(acc, wv, grouped));
}

// Co-operative likelihood evaluation of the proposals S.pw[.][src] of every lane `src` with need set:
// lanes over reads, coalesced 512-byte row reads of the unit's transposed tensor, wave butterfly sum.
// The K*Mh row indices of a request are first computed one per lane and then broadcast with readlane, so the
// global loads depend on nothing but registers.
template <class W>
__device__ inline double coop_eval(bool need, const SimtLdsT<W> &S, const Lane &c, int rpad, int lane) {
  double result = 0.0;
  const int nch = rpad / WAVE;  // wave-uniform number of 64-read chunks
  unsigned long long todo = __ballot(need);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int K = __builtin_amdgcn_readfirstlane(__shfl(c.K, src, WAVE));
    const int Mh = __builtin_amdgcn_readfirstlane(__shfl(c.Mh, src, WAVE));
    const uint32_t amask = (uint32_t)__shfl((int)c.amask, src, WAVE);
    const double invK = __shfl(c.invK, src, WAVE);
    const unsigned long long rtb = __shfl((unsigned long long)(uintptr_t)c.rt, src, WAVE);
    const unsigned long long cwb = __shfl((unsigned long long)(uintptr_t)c.cw, src, WAVE);
    const double *rt = reinterpret_cast<const double *>((uintptr_t)rtb) + lane;
    const double *cw = reinterpret_cast<const double *>((uintptr_t)(cwb & ~1ull)) + lane;
    const bool grouped = (cwb & 1ull) != 0ull;
    double s;
    if (nch == 4) s = coop_body<4, W>(S, src, K, Mh, amask, invK, rt, cw, rpad, lane, grouped);
    else if (nch == 1) s = coop_body<1, W>(S, src, K, Mh, amask, invK, rt, cw, rpad, lane, grouped);
    else if (nch == 2) s = coop_body<2, W>(S, src, K, Mh, amask, invK, rt, cw, rpad, lane, grouped);
    else if (nch == 8) s = coop_body<8, W>(S, src, K, Mh, amask, invK, rt, cw, rpad, lane, grouped);
    else s = coop_body<16, W>(S, src, K, Mh, amask, invK, rt, cw, rpad, lane, grouped);
    if (lane == src) result = s;
  }
  return result;
}

template <class W>
__device__ __forceinline__ uint64_t lane_genotype_tag(const SimtLdsT<W> &S, const Lane &c, int lane) {
  uint64_t t = 0;
  if (c.key_bits * c.K <= 63) {
    for (int h = 0; h < c.K; h++) t = (t << c.key_bits) | (uint64_t)L_(S.pw, h);
  } else {
    for (int h = 0; h < c.K; h++) t = word_mix(t, L_(S.pw, h));
  }
  return (t << 1) | 1ull;
}

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
constexpr int N_STATS = 64;  // [0..23] event counters, [24..35] phase timers of the steps, [36..47] ... of the table completion, [48..55] sub-timers of the steps
static __device__ unsigned long long g_stats[N_STATS];
#endif
#ifdef MCHAP_STATS
#define STAT_ADD(i, pred)                                                         \
  do {                                                                            \
    const int n_ = __popcll(__ballot(pred));                                      \
    if (n_ && (threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)n_); \
  } while (0)
#define STAT_WAVE(i, n)                                                                     \
  do {                                                                                      \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)(n));           \
  } while (0)
#else
#define STAT_ADD(i, pred) \
  do {                    \
  } while (0)
#define STAT_WAVE(i, n) \
  do {                  \
  } while (0)
#endif

// likelihood of the lane's proposal S.pw[.][lane] (where need): cache probe, co-operative evaluation on a miss.
// The per-chain table is 4-way set associative (one 64-byte line per set): a hit in way k > 0 moves the entry one
// way towards the front, a miss inserts into the last way, so the ~40 neighbours a converged chain proposes every
// step stay resident however their keys collide.
template <int KT, class W>
__device__ inline double lane_eval_cached(bool need, const SimtLdsT<W> &S, const Lane &c, int rpad, int lane) {
  double val = 0.0;
  bool miss = need;
  uint64_t tag = 0;
  ulonglong2 *set = nullptr;
  // genotypes wider than 63 bits are tagged by a hash: a tag match is verified against the genotype's words kept beside
  // the entry (c.ckeys), hits do not move entries and a miss replaces the way its tag picks (or the entry it collided with)
  const bool wide = c.key_bits * c.K > 63;
  int wway = 3;
  if (need && c.cache && !wide) {
    tag = lane_genotype_tag(S, c, lane);
    const uint64_t si = mix64(tag) & c.cache_mask;
    set = c.cache + 4 * si;
    const ulonglong2 e0 = set[0], e1 = set[1], e2 = set[2], e3 = set[3];
    if (e0.x == tag) {
      val = __longlong_as_double((long long)e0.y);
      miss = false;
    } else if (e1.x == tag) {
      val = __longlong_as_double((long long)e1.y);
      miss = false;
      set[0] = e1;
      set[1] = e0;
    } else if (e2.x == tag) {
      val = __longlong_as_double((long long)e2.y);
      miss = false;
      set[1] = e2;
      set[2] = e1;
    } else if (e3.x == tag) {
      val = __longlong_as_double((long long)e3.y);
      miss = false;
      set[2] = e3;
      set[3] = e2;
    }
  } else if (need && c.cache && c.ckeys) {
    tag = lane_genotype_tag(S, c, lane);
    const uint64_t si = (tag >> 1) & c.cache_mask;
    set = c.cache + 4 * si;
    wway = (int)((tag >> 40) & 3u);
    int hit = -1;
    for (int w = 3; w >= 0; w--) {
      const ulonglong2 e = set[w];
      if (e.x == tag) hit = w;
      if (e.x == 0ull) wway = w;
    }
    if (hit >= 0) {
      const W *kw = reinterpret_cast<const W *>(c.ckeys + (4 * si + hit) * c.key_words);  // (key_words = K words of W)
      bool same = true;
      for (int h = 0; h < c.K; h++) same = same && (kw[h] == L_(S.pw, h));
      if (same) {
        val = __longlong_as_double((long long)set[hit].y);
        miss = false;
      } else {
        wway = hit;
      }
    }
  }
  STAT_ADD(0, need);
  STAT_ADD(1, miss);
  STAT_ADD(2, true);  // probe rounds x 64 lanes
  const double v = coop_eval(miss, S, c, rpad, lane);
  if (miss) {
    val = v;
    if (set) {
      if (wide) {
        W *kw = reinterpret_cast<W *>(c.ckeys + ((size_t)(set - c.cache) + wway) * c.key_words);
        for (int h = 0; h < c.K; h++) kw[h] = L_(S.pw, h);
      }
      set[wide ? wway : 3] = make_ulonglong2(tag, (unsigned long long)__double_as_longlong(v));
    }
  }
  return val;
}

// mutation.py:14-161 for every lane with act set; all lanes must call (co-operative evaluation inside)
template <int KT, class W>
__device__ inline double simt_base_step(bool act, const SimtLdsT<W> &S, Lane &c, int wb, double llk, int h, int j, double temp,
                                        int amax, int rpad, int lane) {
  const int K = KT ? KT : c.K;
  int n_alleles = 0, sh = 0, current = 0;
  double lhapcount = 0.0, lprior = 0.0;
  W wh = 0;
  if (act) {
    n_alleles = L_(S.nal, j);
    sh = L_(S.shift, j);
    lhapcount = c_ln[lane_count_copies<KT, W>(S.w, wb, K, h, lane)];
    lprior = lane_words_prior<KT, W>(S, c, S.w, wb, lane);
    wh = L_(S.w, wb + h);
    current = (int)((uint32_t)(wh >> sh) & c.amask);
    for (int i = 0; i < K; i++) L_(S.pw, i) = L_(S.w, wb + i);
  }
  double la[MCHAP_MAX_ALLELE], lk[MCHAP_MAX_ALLELE];
  int n_options = 0;
#pragma unroll
  for (int i = 0; i < MCHAP_MAX_ALLELE; i++) {
    la[i] = -INFINITY;
    lk[i] = llk;
    if (i >= amax) continue;  // wave-uniform bound
    const bool prop = act && i < n_alleles && i != current;
    if (prop) {
      n_options += 1;
      L_(S.pw, h) = (wh & ~((W)c.amask << sh)) | ((W)i << sh);
    }
    const double llk_i = lane_eval_cached<KT, W>(prop, S, c, rpad, lane);
    if (prop) {
      lk[i] = llk_i;
      const double llk_ratio = llk_i - llk;
      double lprior_ratio = 0.0;
      if (!isnan(c.inbreeding)) lprior_ratio = lane_words_prior<KT, W>(S, c, S.pw, 0, lane) - lprior;
      const double lproposal_ratio = c_ln[lane_count_copies<KT, W>(S.pw, 0, K, h, lane)] - lhapcount;
      const double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
      la[i] = fmin(0.0, mh);
    }
  }
  if (!act) return llk;
  const double ln_opt = c_ln[n_options];
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < MCHAP_MAX_ALLELE; i++) {
    if (i < n_alleles) {
      la[i] = exp(la[i] - ln_opt);
      sum += la[i];
    }
  }
  const double stay = 1.0 - sum;
  const double u = rng_double(c.rng);
  double cacc = 0.0;
  int choice = n_alleles;
  double llk_new = llk;
#pragma unroll
  for (int i = 0; i < MCHAP_MAX_ALLELE; i++) {
    if (i < n_alleles && choice == n_alleles) {
      cacc += (i == current) ? stay : la[i];
      if (cacc > u) {
        choice = i;
        llk_new = lk[i];
      }
    }
  }
  if (choice >= n_alleles) {
    choice = n_alleles - 1;
#pragma unroll
    for (int i = 0; i < MCHAP_MAX_ALLELE; i++)
      if (i == choice) llk_new = lk[i];
  }
  L_(S.w, wb + h) = (wh & ~((W)c.amask << sh)) | ((W)choice << sh);
  return llk_new;
}

// structural.py:433-587 for every lane with act set
template <int KT, class W>
__device__ inline double simt_interval_step(bool act, const SimtLdsT<W> &S, Lane &c, int wb, double llk, int start, int stop,
                                            int step_type, double temp, int rpad, int lane) {
  const int K = KT ? KT : c.K;
  typedef typename PackOf<W>::type P;
  W min_ = 0;
  P lin = 0, lout = 0;
  // option (h0, h1) of the enumeration -> the `in` label pack after the move (structural.py:121-178 / 240-307)
  auto option_labels = [&](uint8_t hh) -> P {
    const int h0 = hh >> 4, h1 = hh & 15;
    if (step_type == 0) return nib_set(nib_set(lin, h0, nib(lin, h1)), h1, nib(lin, h0));
    return nib_set(lin, h0, nib(lin, h1));
  };
  int n_options = 0;
  double lprior = 0.0, log_proposal_prob = 0.0, ln_opt = 0.0, u = 2.0;
  if (act) {
    const W full = lane_interval_mask<W>(c, 0, c.Mh);
    min_ = lane_interval_mask<W>(c, start, stop);
    lin = lane_segment_labels<KT, W>(S.w, wb, K, min_, lane);
    lout = lane_segment_labels<KT, W>(S.w, wb, K, full & ~min_, lane);
    // enumerate straight into the lane's LDS column
    {
      const P hd = dosage_of_labels(lin, lout, K, true);
      if (step_type == 0) {
        for (int h0 = 0; h0 < K; h0++) {
          if (nib(hd, h0) == 0) continue;
          for (int h1 = h0 + 1; h1 < K; h1++) {
            if (nib(hd, h1) == 0) continue;
            if (nib(lin, h0) == nib(lin, h1) || nib(lout, h0) == nib(lout, h1)) continue;
            L_(S.optin, n_options) = (uint8_t)((h0 << 4) | h1);
            n_options++;
          }
        }
      } else {
        const P sd = dosage_of_labels(lin, lout, K, false);
        for (int h0 = 0; h0 < K; h0++) {
          if (nib(hd, h0) == 0) continue;
          if (nib(sd, h0) == 1) continue;
          for (int h1 = 0; h1 < K; h1++) {
            if (nib(sd, h1) == 0) continue;
            if (nib(lin, h0) == nib(lin, h1)) continue;
            L_(S.optin, n_options) = (uint8_t)((h0 << 4) | h1);
            n_options++;
          }
        }
      }
    }
    if (n_options > 0) {
      log_proposal_prob = c_ln_inv[n_options];
      ln_opt = c_ln[n_options];
      if (!isnan(c.inbreeding)) lprior = lane_words_prior<KT, W>(S, c, S.w, wb, lane);
      u = rng_double(c.rng);  // the only draw of this step; the evaluations below consume none
    }
  }
  const int nmax = wave_max_i(n_options);
  double cacc = 0.0;
  int choice = -1;
  double llk_choice = llk;
  for (int i = 0; i < nmax; i++) {
    const bool prop = act && i < n_options && choice < 0;  // options after the chosen one cannot matter
    P oin = 0;
    if (prop) {
      oin = option_labels(L_(S.optin, i));
      for (int h = 0; h < K; h++) L_(S.pw, h) = (L_(S.w, wb + h) & ~min_) | (L_(S.w, wb + nib(oin, h)) & min_);
    }
    const double llk_i = lane_eval_cached<KT, W>(prop, S, c, rpad, lane);
    if (prop) {
      const double llk_ratio = llk_i - llk;
      double lprior_ratio = 0.0;
      if (!isnan(c.inbreeding)) lprior_ratio = lane_prior_of_dosage(S, c, dosage_of_labels(oin, lout, K, true), lane) - lprior;
      const int n_return = step_type == 0 ? recombination_n_options(oin, lout, K) : dosage_n_options(oin, lout, K);
      const double lproposal_ratio = c_ln_inv[n_return] - log_proposal_prob;
      const double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
      cacc += exp(fmin(0.0, mh) - ln_opt);
      if (cacc > u) {
        choice = i;
        llk_choice = llk_i;
      }
    }
  }
  if (act && choice >= 0) {
    const P oin = option_labels(L_(S.optin, choice));
    for (int h = 0; h < K; h++) L_(S.pw, h) = (L_(S.w, wb + h) & ~min_) | (L_(S.w, wb + nib(oin, h)) & min_);
    for (int h = 0; h < K; h++) L_(S.w, wb + h) = L_(S.pw, h);
    llk = llk_choice;
  }
  return llk;
}

// structural.py:22-71 + 590-673 for every lane with act set.  Returns false for lanes that hit the ValueError.
template <int KT, class W>
__device__ inline bool simt_structural_compound(bool act, const SimtLdsT<W> &S, Lane &c, int wb, double &llk, int n_breaks,
                                                bool whole, int step_type, double temp, int rpad, int lane) {
  bool ok = true;
  W zeros = 0;  // bit i set = interval end point i (n + 1 <= bits of W - 1: 62 / 126 positions)
  int n_int = 0;
  if (act) {
    const int n = c.Mh;
    if (whole) {
      zeros = (W)1 | ((W)1 << n);
      n_int = 1;
    } else if (n_breaks >= n) {
      ok = false;
      act = false;
    } else {
      W ind = 0;
      for (int i = 1; i < n; i++) ind |= (W)1 << i;
      for (int b = 0; b < n_breaks; b++) {
        const int no = word_popc(ind);
        if (no == 0) break;
        int k = (int)rng_interval(c.rng, (uint32_t)(no - 1));
        W t = ind;
        while (k-- > 0) t &= t - 1;
        ind &= ~(t & (~t + 1));
      }
      zeros = ~ind & (((W)1 << (n + 1)) - (W)1);
      n_int = n_breaks + 1;
    }
    if (act) {
      for (int i = 0; i < n_int; i++) L_(S.sub, i) = (uint16_t)i;
      for (int i = n_int - 1; i >= 1; i--) {
        const int k = (int)rng_interval(c.rng, (uint32_t)i);
        const uint16_t a = L_(S.sub, i), b = L_(S.sub, k);
        L_(S.sub, i) = b;
        L_(S.sub, k) = a;
      }
    }
  }
  const int nmax = wave_max_i(act ? n_int : 0);
  for (int i = 0; i < nmax; i++) {
    const bool a2 = act && i < n_int;
    int start = 0, stop = 0;
    if (a2) {
      const int iv = L_(S.sub, i);
      // end points iv and iv+1 = the (iv)-th and (iv+1)-th set bits of zeros
      W z = zeros;
      for (int q = 0; q < iv; q++) z &= z - 1;
      start = word_ffs(z) - 1;
      z &= z - 1;
      stop = word_ffs(z) - 1;
    }
    llk = simt_interval_step<KT, W>(a2, S, c, wb, llk, start, stop, step_type, temp, rpad, lane);
  }
  return ok;
}

template <int KT, class W = uint64_t>
__global__ __launch_bounds__(64) void denovo_simt_kernel(const SimtParams P) {
  constexpr int NW = (int)sizeof(W) / 8;  // uint64 words of the trace per haplotype (most significant first)
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int lane = threadIdx.x;
  const int T = D.n_temps, Cn = D.chains, Sn = D.steps;
  const int Kmax = P.max_ploidy, Mmax = P.max_pos;
  const int rpad = D.rpad;
  // LDS carve (lane-strided arrays)
  SimtLdsT<W> S;
  {
    unsigned char *p = smem;
    S.w = reinterpret_cast<W *>(p); p += sizeof(W) * T * Kmax * 64;
    S.pw = reinterpret_cast<W *>(p); p += sizeof(W) * Kmax * 64;
    S.llk_t = reinterpret_cast<double *>(p); p += (size_t)8 * T * 64;
    S.rngn = reinterpret_cast<uint64_t *>(p); p += (size_t)8 * T * 64;
    S.prior = reinterpret_cast<double *>(p); p += (size_t)8 * (2 * Kmax + 5) * 64;
    S.optin = p; p += (size_t)Kmax * Kmax * 64;
    S.sub = reinterpret_cast<uint16_t *>(p); p += (size_t)2 * Kmax * Mmax * 64;
    S.cols = reinterpret_cast<uint16_t *>(p); p += (size_t)2 * Mmax * 64;
    S.shift = p; p += (size_t)Mmax * 64;
    S.nal = p;
  }
  const long long q = (long long)blockIdx.x * WAVE + lane;  // global chain index
  const long long n_chains = (long long)P.n_units * Cn;
  Lane c;
  c.alive = q < n_chains;
  const int u = c.alive ? (int)(q / Cn) : 0;
  const int chain = c.alive ? (int)(q % Cn) : 0;
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);
  if (c.alive && mi[META_I_STATUS] != MCHAP_UNIT_OK) c.alive = false;
  const int A = U.max_allele;
  c.K = c.alive ? U.ploidy : 1;
  c.Mh = c.alive ? mi[META_I_MH] : 1;
  c.bits = allele_bits(A);
  c.amask = (1u << c.bits) - 1u;
  c.invK = 1.0 / (double)c.K;
  c.inbreeding = U.inbreeding;
  c.key_bits = c.bits * c.Mh;
  c.rt = P.rt + (size_t)u * P.max_ma * rpad;
  c.cw = reinterpret_cast<const double *>((uintptr_t)(P.cntw + (size_t)u * rpad) | (uintptr_t)((c.alive && mi[META_I_W01] != 0) ? 1 : 0));
  c.cache = nullptr;
  c.cache_mask = 0;
  c.ckeys = nullptr;
  c.key_words = D.cache_key_words;
  if (D.cache_slots > 0) {
    c.cache = reinterpret_cast<ulonglong2 *>(D.cache) + (size_t)q * (size_t)D.cache_slots;
    if (D.cache_keys) c.ckeys = D.cache_keys + (size_t)q * (size_t)D.cache_slots * D.cache_key_words;
    c.cache_mask = (uint32_t)(D.cache_slots / 4) - 1u;  // sets of 4 ways
  }
  const int K = KT ? KT : c.K;
  const int Mh = c.Mh;
  for (int j = 0; j < Mh; j++) {
    L_(S.cols, j) = (uint16_t)(c.alive ? mi[META_I_COLS + j] : 0);
    L_(S.nal, j) = (uint8_t)(c.alive ? mi[META_I_COLS + P.max_pos + j] : 2);
    L_(S.shift, j) = (uint8_t)(c.bits * (Mh - 1 - j));
  }
  if (c.alive && !isnan(c.inbreeding))
    for (int i = 0; i < 2 * K + 5; i++) L_(S.prior, i) = mf[meta_f_prior(0) + i];
  // wave-uniform loop bounds
  const int amax = wave_max_i(c.alive ? A : 0);
  // ---- initial genotype (assemble/mcmc.py:202-208) into temperature slot 0 ----
  if (c.alive) {
    if (U.initial_off >= 0) {
      const int8_t *ini = D.initial + U.initial_off + (size_t)chain * K * Mh;
      for (int h = 0; h < K; h++) {
        W x = 0;
        for (int j = 0; j < Mh; j++) x |= (W)(uint8_t)ini[h * Mh + j] << L_(S.shift, j);
        L_(S.w, h) = x;
      }
    } else {
      const double *dist = mf + meta_f_dist(P.max_ploidy);
      rng_open(c.rng, D.seed, U.stream_id, (uint32_t)chain, SLOT_INIT, 0);
      for (int h = 0; h < K; h++) {
        W x = 0;
        for (int j = 0; j < Mh; j++) {
          double s = 0.0;
          for (int a = 0; a < A; a++) s += dist[j * A + a];
          double cacc = 0.0;
          const double uu = rng_double(c.rng);
          int ch = A;
          for (int a = 0; a < A; a++) {
            cacc += dist[j * A + a] / s;
            if (cacc > uu) {
              ch = a;
              break;
            }
          }
          if (ch >= A) ch = A - 1;
          x |= (W)ch << L_(S.shift, j);
        }
        L_(S.w, h) = x;
      }
    }
    for (int h = 0; h < K; h++) L_(S.pw, h) = L_(S.w, h);
  }
  {
    const double llk0 = coop_eval(c.alive, S, c, rpad, lane);  // assemble/mcmc.py:303 (not cached there either)
    if (c.alive) {
      for (int t = T - 1; t >= 0; t--) {
        for (int h = 0; h < K; h++) L_(S.w, t * Kmax + h) = L_(S.w, h);
        L_(S.llk_t, t) = llk0;
        L_(S.rngn, t) = 0;
      }
    }
  }
  const double *break_dist = D.break_table + (size_t)Mh * D.max_pos;
  const int n_break_dist = D.n_intervals > 0 ? D.n_intervals : Mh;
  const size_t trace_base = U.trace_off + (size_t)chain * Sn * K * NW;
  const size_t llk_base = U.llk_off + (size_t)chain * Sn;
  int status = MCHAP_UNIT_OK;
  const int nsub = K * Mh;

  for (int step = 0; step < Sn; step++) {
    for (int t = 0; t < T; t++) {
      const int wb = t * Kmax;
      double llk = 0.0;
      const double temp = D.temps[t];
      if (c.alive) {
        llk = L_(S.llk_t, t);
        if (isnan(llk)) {  // assemble/mcmc.py:330-331
          status = MCHAP_UNIT_NAN_LLK;
          c.alive = false;
        }
      }
      if (c.alive) {
        rng_open(c.rng, D.seed, U.stream_id, (uint32_t)chain, (uint32_t)t, (uint64_t)step * STEP_DRAWS);
        // mutation.compound_step: shuffle (mutation.py:219-229)
        // entry i = (h << 8) | j of sub-step h * Mh + j (no division when it is consumed)
        for (int h = 0, i = 0; h < K; h++)
          for (int j = 0; j < Mh; j++, i++) L_(S.sub, i) = (uint16_t)((h << 8) | j);
        for (int i = nsub - 1; i >= 1; i--) {
          const int k = (int)rng_interval(c.rng, (uint32_t)i);
          const uint16_t a = L_(S.sub, i), b = L_(S.sub, k);
          L_(S.sub, i) = b;
          L_(S.sub, k) = a;
        }
      }
      const int nsub_max = wave_max_i(c.alive ? nsub : 0);
      for (int i = 0; i < nsub_max; i++) {
        const bool act = c.alive && i < nsub;
        int h = 0, j = 0;
        if (act) {
          const int s = L_(S.sub, i);
          h = s >> 8;
          j = s & 255;
        }
        llk = simt_base_step<KT, W>(act, S, c, wb, llk, h, j, temp, amax, rpad, lane);
      }
      for (int kind = 0; kind < 3; kind++) {
        bool act = false;
        int nb = 0;
        if (c.alive) {
          const double pstep = kind == 0 ? D.p_recomb : (kind == 1 ? D.p_partial : D.p_dosage);
          act = rng_double(c.rng) <= pstep;
          if (act && kind < 2) {
            if (D.n_intervals > 0) {
              (void)rng_double(c.rng);
              nb = D.n_intervals - 1;
            } else {
              nb = choose_from(break_dist, n_break_dist, rng_double(c.rng));
            }
          }
        }
        const bool ok = simt_structural_compound<KT, W>(act, S, c, wb, llk, nb, kind == 2, kind == 2 ? 1 : kind, temp, rpad, lane);
        if (!ok) {
          status = MCHAP_UNIT_BREAKS;
          c.alive = false;
        }
      }
      if (c.alive) {
        if (t > 0) {
          // tempering.py:61-151
          const int wj = (t - 1) * Kmax;
          double llk_j = L_(S.llk_t, t - 1);
          const double prior_i = lane_words_prior<KT, W>(S, c, S.w, wb, lane);
          const double prior_j = lane_words_prior<KT, W>(S, c, S.w, wj, lane);
          const double ui = llk + prior_i, uj = llk_j + prior_j;
          double acc = exp((uj - ui) * temp + (ui - uj) * D.temps[t - 1]);
          if (acc > 1.0) acc = 1.0;
          const double val = rng_double(c.rng);
          if (acc >= val) {
            for (int h = 0; h < K; h++) {
              const W x = L_(S.w, wb + h);
              L_(S.w, wb + h) = L_(S.w, wj + h);
              L_(S.w, wj + h) = x;
            }
            const double x = llk;
            llk = llk_j;
            llk_j = x;
          }
          L_(S.llk_t, t - 1) = llk_j;
        }
        L_(S.llk_t, t) = llk;
        L_(S.rngn, t) = c.rng.n;
      }
    }
    if (c.alive) {
      // record the cold chain with its haplotypes in canonical order (assemble/classes.py:265-278)
      const int wb = (T - 1) * Kmax;
      for (int h = 0; h < K; h++) {
        const W x = L_(S.w, wb + h);
        int rank = 0;
        for (int g = 0; g < K; g++) {
          const W y = L_(S.w, wb + g);
          rank += (y < x || (y == x && g < h)) ? 1 : 0;
        }
        if (NW == 1) {
          D.trace[trace_base + (size_t)step * K + rank] = (uint64_t)x;
        } else {
          D.trace[trace_base + ((size_t)step * K + rank) * 2] = (uint64_t)((u128)x >> 64);
          D.trace[trace_base + ((size_t)step * K + rank) * 2 + 1] = (uint64_t)x;
        }
      }
      D.llks[llk_base + step] = L_(S.llk_t, T - 1);
    }
  }
  if (status != MCHAP_UNIT_OK) atomicMax(&D.status[u], status);
}

#undef L_

}  // namespace mchap
