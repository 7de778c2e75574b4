// De-novo MCMC sampler, speculative form for MI355X (gfx950): a GROUP of G lanes per chain (G = 16/32/64).
//
// The reference's step is a chain of ~40 dependent sub-steps (assemble/mutation.py:164-246,
// assemble/structural.py:590-673): each proposes a few neighbours of the current genotype, accepts one with small
// probability, and hands the (usually unchanged) genotype to the next.  Two facts make the chain parallel:
//   * the random numbers are counter-based (philox.hpp): the uniform a sub-step will consume is known from its
//     position alone, before any earlier sub-step has run;
//   * a sub-step's proposal probabilities depend only on the current genotype.
// So all pending sub-steps of a compound step are evaluated at once, one per lane, against the current genotype
// (cache probe per lane, co-operative likelihood evaluation for the misses); the first sub-step in sequence order
// that moves is applied, everything before it is thereby validated as "stay", and only the sub-steps after it are
// evaluated again.  The result is bit-for-bit the sequential chain (same draws, same decisions); the number of
// parallel rounds per compound step is 1 + the number of accepted moves instead of the number of sub-steps.
// The Fisher-Yates shuffle of the sub-steps is done by every lane tracing its own element through the n-1
// transpositions (draws computed in parallel).  Structural steps speculate over (interval, option) pairs the same
// way; their draws are data dependent (structural.py:504-506) and are consumed in the sequential validation walk.
//
// On top of that (all results-neutral): memoised no-move outcomes per genotype (bounds of the
// mutation step's uniforms, total move probability of every visited interval step), a window of staged draws shared
// by the compound steps of one MCMC step, the read table as uint8 codes into a per-unit dictionary held in LDS, and
// reuse of the current genotype's haplotype products by the requests of one chain.
//
// Launch: one wavefront per 64/G chains, all units of a launch share the ploidy KT; prepare kernel and workspace
// as for the lanes-over-chains kernel (denovo_simt_kernel.hpp).  Same traces as the other two kernels.
#pragma once
#include <type_traits>

#include "denovo_simt_kernel.hpp"

#ifdef MCHAP_MASK_PADDING
#define MCHAP_PAD_LANE(x) (x)
#else
#define MCHAP_PAD_LANE(x) true
#endif
#ifndef MCHAP_COOP_UNR
#define MCHAP_COOP_UNR 2  // row loads in flight per lane and chunk
#endif
#ifndef MCHAP_SPEC_TB
#define MCHAP_SPEC_TB 4    // trace records per flush
#endif
#ifndef MCHAP_SPEC_LOW
#define MCHAP_SPEC_LOW 8   // staged draws a structural step wants to find before it refills the window
#endif
#ifndef MCHAP_SPEC_CG
#define MCHAP_SPEC_CG 4  // pairs whose dictionary gathers are in flight together in spec_coop_coded (ploidy > 4)
#endif
// Instantiation variants (MCHAP_SPEC_VAR, set per object by spec_inst.hip; the kernel's last template argument names them):
//   0 plain: what a batch of units with 65 to 256 reads needs (configs[1]) and nothing else;
//   1 "side by side" (MCHAP_SPEC_SBS): several requests of a unit of at most 64 reads evaluated in one pass of the wavefront
//     (spec_coop_all) -- for batches that hold such a unit;
//   2 "deep" (MCHAP_SPEC_DEEP): haplotype products of the current genotype kept in the workspace for the read chunks beyond
//     the first four, and product reuse for up to 192 (haplotype, position) pairs -- for batches with more than 256 reads per
//     unit or more than 128 pairs (configs[4]).
// They are separate instantiations because the extra paths cost registers: compiled into one kernel they took 1-4 % off the
// evaluation of configs[1] (spills in the kernel's hot region: 48 -> 88-97 spilled VGPRs).  Same traces whichever runs.
#ifndef MCHAP_SPEC_VAR
#define MCHAP_SPEC_VAR 0
#endif
#define MCHAP_SPEC_SBS (MCHAP_SPEC_VAR == 1)
#define MCHAP_SPEC_DEEP (MCHAP_SPEC_VAR == 2)
#ifndef MCHAP_SPEC_WIN0
#define MCHAP_SPEC_WIN0 4     // mutation sub-steps a chain's first compound step speculates over per round
#endif
#ifndef MCHAP_SPEC_WIN_MIN
#define MCHAP_SPEC_WIN_MIN 2  // ... and the least a converging chain falls back to
#endif
#ifndef MCHAP_REUSE_MAXK
#define MCHAP_REUSE_MAXK 8  // largest ploidy whose kernels carry the product-reuse path (K x 4 products in registers)
#endif
#ifndef MCHAP_CODED_UNR
#define MCHAP_CODED_UNR 16  // code loads (one register each) in flight per lane in the coded evaluation
#endif

namespace mchap {

constexpr int SPEC_MAX_IV = 64;  // intervals per structural compound step (<= n_pos)
constexpr int SPEC_TB = MCHAP_SPEC_TB;    // trace records per flush: K = 4 -> one 128-byte line of words + 32 bytes of llks
constexpr int SPEC_LN = 72;   // log tables: counts up to K(K-1) <= 56
constexpr int SPEC_DRAWS_MAX = 384;  // draws of the current stream staged in LDS per group (Philox blocks in parallel)
__host__ __device__ inline int spec_draws(int K, int Mmax) {
  int d = 2 * K * Mmax - 1;
  if (d < 3 * Mmax + 2) d = 3 * Mmax + 2;
  d = (d + 1) & ~1;
  return d < SPEC_DRAWS_MAX ? d : SPEC_DRAWS_MAX;
}

// LDS pointers carry their address space so that every access is a ds_* instruction (a pointer stored in a struct
// otherwise degrades to a generic "flat" access)
#define LDSP(T) __attribute__((address_space(3))) T *
#define GLBP(T) __attribute__((address_space(1))) T *
template <class T>
__device__ __forceinline__ LDSP(T) lds_cast(void *p) {
  return (LDSP(T))p;
}

// The unit's coded table and read weights: in HBM / L2 (LT = false, the speculative kernel) or copied into the wave's LDS
// (LT = true, the settling kernel of denovo_lane_kernel.hpp, whose evaluations are then free of memory latency).
template <bool LT>
struct TabPtr;
template <>
struct TabPtr<false> {
  typedef GLBP(const uint8_t) u8;
  typedef GLBP(const double) f64;
  template <class CT>
  static __device__ __forceinline__ CT ld(u8 p) { return *reinterpret_cast<GLBP(const CT)>(p); }
};
template <>
struct TabPtr<true> {
  typedef LDSP(const uint8_t) u8;
  typedef LDSP(const double) f64;
  template <class CT>
  static __device__ __forceinline__ CT ld(u8 p) { return *reinterpret_cast<LDSP(const CT)>(p); }
};

enum { GP_RT = 0, GP_CW, GP_CT, GP_CACHE, GP_CKEYS, GP_TRACE, GP_LLK, GP_GBP, GP_N };
enum { GV_INB = 0, GV_MLO, GV_MHI, GV_N };
struct SpecLds {
  LDSP(uint64_t) pw;      // [K][64] request words of missing lanes (lane strided)
  LDSP(uint64_t) wst;     // [NG][T][K] genotype of every temperature
  LDSP(double) llk_t;     // [NG][T]
  LDSP(uint64_t) rngn;    // [NG][T]
  LDSP(double) prior;     // [NG][2K+5]
  LDSP(double) ptab;      // [64] per-slot move probability
  LDSP(double) optp;      // [MCHAP_MAX_ALLELE][64] mutation option probabilities of the lane
  LDSP(double) optl;      // [MCHAP_MAX_ALLELE][64] ... and their log likelihoods
  LDSP(double) ln;        // [SPEC_LN] log(n)
  LDSP(double) lninv;     // [SPEC_LN] log(1/n)
  LDSP(uint32_t) ivse;    // [NG][SPEC_MAX_IV] start | stop << 8, in visiting order
  LDSP(uint32_t) ivlin;   // [NG][SPEC_MAX_IV]
  LDSP(uint32_t) ivlout;  // [NG][SPEC_MAX_IV]
  LDSP(uint32_t) ivno;    // [NG][SPEC_MAX_IV] n_options
  LDSP(uint16_t) cols;    // [NG][Mmax]
  LDSP(uint16_t) permtab; // [NG][nmax] (h << 8) | j by order position
  LDSP(uint8_t) shift;    // [NG][Mmax]
  LDSP(uint8_t) nal;      // [NG][Mmax]
  LDSP(uint8_t) ktab;     // [NG][nmax]
  LDSP(uint8_t) ordtab;   // [NG][SPEC_MAX_IV]
  LDSP(uint16_t) nreads;  // [NG] reads of the group's unit (lanes beyond it do not load the table's padding)
  LDSP(uint16_t) ndict;   // [NG] entries of the unit's dictionary (0: no coded table, float64 rows are read)
  LDSP(double) dict;      // [NG][DICT_MAX] the unit's distinct table values
  LDSP(uint64_t) gptr;    // [NG][GP_N] cold per-chain pointers (kept out of the registers): see GP_*
  LDSP(double) gval;      // [NG][GV_N] cold per-chain values: inbreeding, mutation memo bounds
  LDSP(uint32_t) gstream; // [NG][4] Philox key and counter words of the chain's current stream (Stream)
  LDSP(uint64_t) bw;      // [NG][K] the chain's current genotype while its proposals are evaluated (base words)
  LDSP(double) bpc;       // [K][4][64] one chain per wave only (G = 64), else null: the haplotype products of ...
  LDSP(uint64_t) bpt;     // [K + 1] ... these base words (bpt[K] != 0: valid), kept from one evaluation call to the next
  LDSP(uint64_t) gbt;     // [NG][K + 1] the words whose products the chain's rows of SimtParams::gbp hold (deep units), [K] != 0: valid
  LDSP(const uint8_t) sct;  // code of read `lane` in every row of the table (row r at sct_off(r)), for a unit of at most 64 reads
                            // and 24 K rows in a one-chain-per-wave launch: kept in the product cache's unused chunk slots; else null
  LDSP(uint64_t) tbuf;    // [NG][SPEC_TB][K + 1] trace records (K sorted words + llk) waiting to be written as a line
  LDSP(uint64_t) lc;      // [lc_mask + 1][2] {tag, value bits}: the chain's likelihood cache IN LDS (one chain per wave), or lc_mask == 0
  uint32_t lc_mask;       // entries - 1 of that front cache (0: none -- the cache in the workspace is probed instead)
  uint32_t cache_mask;    // sets of the likelihood cache - 1; cache off: cache_on == false
  int key_words;          // words per entry of the wide-genotype key table (DenovoParams::cache_key_words)
  bool cache_on;
  uint64_t epoch = 0ull;  // SimtParams::cache_epoch (this call's epoch << 33, part of every tag of packed genotypes; 0: none)
  bool reuse_on;          // haplotype products of the chain's current genotype are reused by its proposals
  bool cut_off = false;   // tuning flag 262144: spec_mutation evaluates the misses behind a sub-step known to move as well
  LDSP(const uint8_t) lds_ct;  // the unit's coded table / read weights copied into LDS (settling kernel), else null
  LDSP(const double) lds_cw;
  int crow;               // bytes per row of the coded table (64 lanes x SimtParams::cstride)
  LDSP(uint64_t) draws;   // [NG][SPEC_DRAWS] the two 32-bit words (lo, hi) of draws base .. base + SPEC_DRAWS - 1
  LDSP(double) memo_tot;     // [NG][2][(Mmax+1)^2] total move probability of an interval step for the current
                             // genotype; NaN = not evaluated yet, -1 = the step has no options
  LDSP(double) bdist;        // [NG][Mmax] the chain's cumulative break-count distribution
  int memo_stride;           // 2 * spec_memo_entries(Mmax), or 0 when the tables do not fit
  int ndraws;                // staged draws per group
  // Decision contexts per genotype (CTX instantiations, one chain per wave; see "decision contexts" below): ONE pointer to their
  // LDS area -- [SPEC_CTX_MAX] uint64 directory (tag of the genotype whose context sits in slot i of the chain's region; 0: empty),
  // [SPEC_CTX_MAX] uint32 last use of slot i (the least recently used slot is replaced), [CX_N] uint32 the chain's context state
  // (below), [K Mmax] float64 the CURRENT genotype's move probabilities by sub-step e = h Mh + j (-1: not known yet), [K Mmax]
  // float64 the likelihoods of the genotypes those moves lead to -- read through the CXP_* accessors (more pointers here would
  // be more registers held through the whole kernel)
  LDSP(unsigned char) cx_lds;   // directory, last uses, state
  LDSP(double) cx_val;          // move probabilities, then likelihoods
  uint64_t *cx_base = nullptr;  // the chain's region of the workspace (SimtParams::ctx), or null: no contexts in this launch
  int cx_n = 0;                 // slots in use
};

// interval-step memo (see spec_structural): only for a single temperature and while it stays small
// entries of one interval-step memo table: intervals (start, stop) with 0 <= start < stop <= Mmax, triangular
__host__ __device__ inline int spec_memo_entries(int Mmax) { return Mmax * (Mmax + 1) / 2; }
__host__ __device__ inline int spec_memo_index(int start, int stop) { return stop * (stop - 1) / 2 + start; }
__host__ __device__ inline size_t spec_memo_bytes(int Mmax, int T, int G) {
  const size_t per_group = (size_t)2 * spec_memo_entries(Mmax) * 8;
  if (T != 1 || per_group > 16 * 1024) return 0;
  return per_group * (64 / G);
}

constexpr int SPEC_LC_ENTRIES = 256;  // entries of the LDS front cache of a chain's likelihoods (16 bytes each)
#ifndef MCHAP_LC_GEN
#define MCHAP_LC_GEN 48
#endif
constexpr uint32_t SPEC_LC_SECOND_LEVEL_GEN = MCHAP_LC_GEN;  // genotype changes after which a front-cache miss also probes the workspace table
__host__ __device__ inline size_t spec_lc_bytes(int entries = SPEC_LC_ENTRIES) { return (size_t)16 * entries + 16; }
// SimtParams::bp_cache: bit 0 the base-product cache, bit 1 the front cache, bit 2 ... with half the entries (the launches with
// decision contexts where the full table would cost a resident wavefront per CU)
// ... bit 3: every unit of the batch has at most 64 reads and a small table, so the product cache's chunk slots 1..3 of its last two
// haplotypes are free (sct_off: a one-chunk unit's code table starts at haplotype 0's) -- the decision contexts' LDS goes there
// instead of behind the front cache: their directory into haplotype K - 1's slots, their values into haplotype K - 2's
__host__ __device__ inline bool spec_ctx_in_bpc(int K, int Mmax, int Amax, int max_reads) {
  return K >= 3 && max_reads <= 64 && Mmax * Amax <= 24 * (K - 2) && 16 * K * Mmax <= 3 * 64 * 8 && 12 * 64 + 4 * 8 <= 3 * 64 * 8;
}
__host__ __device__ inline int spec_lc_entries(int bp_cache) { return (bp_cache & 2) ? ((bp_cache & 4) ? SPEC_LC_ENTRIES / 2 : SPEC_LC_ENTRIES) : 0; }
// Where row `row` of a one-chunk unit's code table sits behind SpecLds::sct: the product cache is [K][4 chunks][64] float64 and a
// one-chunk unit only uses chunk 0 of every haplotype, so the chunks 1..3 of haplotype h hold the rows 24 h .. 24 h + 23 (64
// codes each) -- 24 K rows in all (round 4, late: haplotype 0's slots alone held 24 rows, i.e. 12 biallelic SNVs; docs/example's
// units that keep moving have 13 to 23)
__device__ __forceinline__ uint32_t sct_off(uint32_t row) { return (row / 24u) * (uint32_t)(4 * WAVE * 8) + (row % 24u) * (uint32_t)WAVE; }
// LDS of the base-product cache (SpecLds::bpc / bpt) of a one-chain-per-wave launch
__host__ __device__ inline size_t spec_bp_cache_bytes(int K) { return (size_t)8 * K * 4 * 64 + (size_t)8 * (K + 1); }

// dynamic LDS of denovo_coast_kernel<NW> (denovo_coast_kernel.hpp): a chain's two memo tables, its break distribution,
// its sorted words, the wavefronts' first undecided steps
constexpr int COAST_NW_LIST = 4;  // wavefronts per chain of a coasting launch over a list of handed-back chains
__host__ __device__ inline size_t coast_lds_bytes(int Mmax) {
  return (size_t)8 * (2 * spec_memo_entries(Mmax) + Mmax) + (size_t)8 * 12;
}

__host__ __device__ inline size_t spec_lds_bytes(int K, int Mmax, int Amax, int T, int G) {
  const int NG = 64 / G;
  const int nmax = K * Mmax;
  const int nopt = Amax > 1 ? Amax - 1 : 1;
  const int niv = Mmax + 1;
  size_t b = 0;
  b += (size_t)8 * K * 64;               // pw
  b += (size_t)8 * NG * T * K;           // wst
  b += (size_t)8 * NG * T * 2;           // llk_t, rngn
  b += (size_t)8 * NG * (2 * K + 5);     // prior
  b += (size_t)8 * 64;                   // ptab
  b += (size_t)8 * nopt * 64 * 2;        // optp, optl
  b += (size_t)8 * SPEC_LN * 2;          // ln, lninv
  b += (size_t)8 * NG * Mmax;            // break distribution row
  b += (size_t)4 * NG * niv * 4;         // ivse, ivlin, ivlout, ivno
  b += (size_t)2 * NG * Mmax;            // cols
  b += (size_t)2 * NG * nmax;            // permtab
  b += (size_t)NG * Mmax * 2;            // shift, nal
  b += (size_t)NG * nmax;                // ktab
  b += (size_t)NG * niv;                 // ordtab
  b = (b + 1) & ~(size_t)1;
  b += (size_t)2 * NG * 2;               // nreads, ndict
  b = (b + 15) & ~(size_t)15;
  b += (size_t)8 * NG * DICT_MAX;        // dict
  b += (size_t)8 * NG * GP_N;            // gptr
  b += (size_t)8 * NG * GV_N;            // gval
  b = (b + 15) & ~(size_t)15;
  b += (size_t)16 * NG;                  // gstream
  b += (size_t)8 * NG * K;               // bw
  b += (size_t)8 * NG * (K + 1);         // gbt
  b += (size_t)8 * NG * SPEC_TB * (K + 1);   // tbuf
  b = (b + 15) & ~(size_t)15;
  b += (size_t)8 * NG * spec_draws(K, Mmax);
  b += spec_memo_bytes(Mmax, T, G);
  return (b + 63) & ~(size_t)63;
}

// stateless Philox draws of a stream (same contract as Rng in philox.hpp)
struct Stream {
  uint32_t k0, k1, c2, c3;
};
__device__ __forceinline__ void stream_words(const Stream &s, uint64_t n, uint32_t &a, uint32_t &b) {
  uint32_t o[4];
  const uint64_t blk = n >> 1;
  philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s.c2, s.c3, s.k0, s.k1, o);
  a = (n & 1) ? o[2] : o[0];
  b = (n & 1) ? o[3] : o[1];
}
__device__ __forceinline__ double stream_double(const Stream &s, uint64_t n) {
  uint32_t a, b;
  stream_words(s, n, a, b);
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ uint32_t stream_interval(const Stream &s, uint64_t n, uint32_t max) {
  uint32_t a, b;
  stream_words(s, n, a, b);
  return __umulhi(a, max + 1u);
}

// Stage draws base .. base + count - 1 (count <= SPEC_DRAWS) of stream `s` into the group's LDS table: every lane
// computes whole Philox blocks (two draws each), so a compound step costs count / (2 G) blocks per lane instead of
// one block per draw.  Entry i holds the two words of draw base + i.
template <int G>
__device__ __forceinline__ void stage_draws(const Stream &s, uint64_t base, int count, LDSP(uint64_t) tab, int gl, bool active) {
  if (active) {
    const uint64_t b0 = base >> 1;
    const int nblk = (int)(((base + (uint64_t)count + 1) >> 1) - b0);
    for (int b = gl; b < nblk; b += G) {
      uint32_t o[4];
      const uint64_t blk = b0 + (uint64_t)b;
      philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s.c2, s.c3, s.k0, s.k1, o);
      const long long i0 = (long long)(blk << 1) - (long long)base;  // table index of the block's first draw
      if (i0 >= 0 && i0 < count) tab[i0] = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
      if (i0 + 1 >= 0 && i0 + 1 < count) tab[i0 + 1] = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
    }
  }
}
__device__ __forceinline__ double draw_double(uint64_t w) {
  return ((double)((uint32_t)w >> 5) * 67108864.0 + (double)((uint32_t)(w >> 32) >> 6)) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ uint32_t draw_interval(uint64_t w, uint32_t max) { return __umulhi((uint32_t)w, max + 1u); }

template <int G>
__device__ __forceinline__ uint64_t grp_ballot(bool p, int gi) {
  const unsigned long long m = __ballot(p);
  if (G == 64) return m;
  return (m >> (gi * G)) & ((1ull << G) - 1ull);
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// max over the G lanes of a group, in every lane.  Within a row of 16 lanes the partners come through DPP row
// rotations (no LDS crossbar round trip per step, unlike __shfl_xor); max is exact, so the order does not matter.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int G>
__device__ __forceinline__ double grp_max_f64(double v) {
#pragma unroll
  for (int o = G / 2; o >= 16; o >>= 1) v = fmax(v, __shfl_xor(v, o, G));
  v = fmax(v, dpp_f64<0x128>(v));  // row_ror:8
  v = fmax(v, dpp_f64<0x124>(v));  // row_ror:4
  v = fmax(v, dpp_f64<0x122>(v));  // row_ror:2
  v = fmax(v, dpp_f64<0x121>(v));  // row_ror:1
  return v;
}
__device__ __forceinline__ void lds_sync() {
#ifdef MCHAP_SYNC_BARRIER
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#endif
}

// The K haplotype words of a genotype.  The helpers take it BY VALUE and the kernel copies the chain's words into
// a local before looping over them: an index that is only constant after loop unrolling, applied directly to a
// member of the chain context (Grp), keeps the whole context in scratch memory (the early SROA pass runs before
// unrolling, and instcombine then folds the selects over its loads into address arithmetic for good).
template <int KT>
struct GWords {
  uint64_t w[KT];
};

template <int KT>
__device__ __forceinline__ uint64_t sel_word(const GWords<KT> g, int h) {
  uint64_t x = g.w[0];
#pragma unroll
  for (int i = 1; i < KT; i++) x = (h == i) ? g.w[i] : x;
  return x;
}
template <int KT>
__device__ __forceinline__ void set_word(GWords<KT> &gref, int h, uint64_t v) {
  GWords<KT> g = gref;
#pragma unroll
  for (int i = 0; i < KT; i++) g.w[i] = (h == i) ? v : g.w[i];
  gref = g;
}
template <int KT>
__device__ __forceinline__ int copies_of(const GWords<KT> g, uint64_t x) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < KT; i++) n += (g.w[i] == x) ? 1 : 0;
  return n;
}
template <int KT>
__device__ __forceinline__ uint32_t dosage_words(const GWords<KT> g) {
  uint32_t d = 0;
#pragma unroll
  for (int h = 0; h < KT; h++) d |= 1u << (4 * h);
#pragma unroll
  for (int h = 0; h < KT; h++) {
    if (nib(d, h) == 0) continue;
#pragma unroll
    for (int p = h + 1; p < KT; p++) {
      if (nib(d, p) == 0) continue;
      if (g.w[h] == g.w[p]) {
        d += 1u << (4 * h);
        d &= ~(15u << (4 * p));
      }
    }
  }
  return d;
}
template <int KT>
__device__ __forceinline__ uint32_t seg_labels(const GWords<KT> g, uint64_t mask) {
  uint32_t lab = 0;
#pragma unroll
  for (int h = 1; h < KT; h++) {
    int l = h;
#pragma unroll
    for (int q = KT - 1; q >= 0; q--)
      if (q < h && ((g.w[q] ^ g.w[h]) & mask) == 0) l = q;
    lab |= (uint32_t)l << (4 * h);
  }
  return lab;
}
template <int KT>
__device__ __forceinline__ double prior_of(LDSP(double) pt, double inbreeding, uint32_t d) {
  if (inbreeding == 0.0) {
    double den = 0.0;
#pragma unroll
    for (int i = 0; i < KT; i++) den += pt[KT + 1 + nib(d, i)];
    return (pt[2 * KT + 3] - den) - pt[2 * KT + 4];
  }
  double prod = 0.0;
#pragma unroll
  for (int i = 0; i < KT; i++) {
    const uint32_t dose = nib(d, i);
    if (dose > 0) prod += pt[dose];
  }
  return pt[2 * KT + 2] + prod;
}
// epoch: SimtParams::cache_epoch -- with it the packed genotype is at most 32 bits (the host's condition) and the call's epoch sits
// in the tag's upper 31 bits: what an earlier call left in the table never matches, so the table is not cleared between calls
template <int KT>
__device__ __forceinline__ uint64_t tag_of(const GWords<KT> g, int key_bits, uint64_t epoch = 0ull) {
  uint64_t t = 0;
  if (key_bits * KT <= 63) {
#pragma unroll
    for (int h = 0; h < KT; h++) t = (t << key_bits) | g.w[h];
    return (t << 1) | 1ull | epoch;
  } else {
#pragma unroll
    for (int h = 0; h < KT; h++) t = mix64(t ^ g.w[h]) + 0x9E3779B97F4A7C15ull;
  }
  return (t << 1) | 1ull;
}
__device__ __forceinline__ uint64_t mask_of(int bits, int Mh, int start, int stop) {
  const int nb = bits * (stop - start);
  const uint64_t ones = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
  const int sh = bits * (Mh - stop);
  return sh >= 64 ? 0ull : ones << sh;
}

// per-group chain context (identical in every lane of the group)
template <int KT>
struct Grp {
  int Mh, bits;
  bool alive;
  bool flat;     // the unit's table holds gaps only (META_I_FLAT): every genotype has the chain's current likelihood
  uint64_t ctr;  // next draw of the current stream (Philox words of the stream: SpecLds::gstream)
  int doff, dcount;  // staged window of the group's draw table: entry doff holds draw ctr, dcount entries are valid
  double llk;
  GWords<KT> g;  // genotype of the current temperature
  // memo of the current genotype (single temperature only): a mutation compound step moves nothing if every one
  // of its uniforms u satisfies mlo <= u < mhi; gen tags the interval-step memo entries
  bool memo_on, mvalid;  // bounds mlo / mhi: SpecLds::gval
  int mwin;              // sub-steps the next mutation compound step speculates over per round (spec_mutation)
  int fill_part, fill_parts;  // PIPE_FILLONLY: this group completes the unknown entries e with e % fill_parts == fill_part
  uint32_t gen, memo_gen;  // memo_gen: the generation the interval memo table currently describes
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
  unsigned long long ph[20], pt0;
#endif
};
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
#define GPHASE(c, i)                                              \
  do {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
    (c).ph[i] += t_ - (c).pt0;                                    \
    (c).pt0 = t_;                                                 \
  } while (0)
#define GSUB_T0() const unsigned long long gsub_t0_ = __builtin_amdgcn_s_memtime()
#define GSUB(c, i) const_cast<unsigned long long &>((c).ph[i]) += __builtin_amdgcn_s_memtime() - gsub_t0_
#define GCOUNT(c, i, n) const_cast<unsigned long long &>((c).ph[i]) += (n)
#define GT0(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#define GT1(c, i, name) const_cast<unsigned long long &>((c).ph[i]) += __builtin_amdgcn_s_memtime() - (name)
#else
#define GT0(name)
#define GT1(c, i, name)
#define GPHASE(c, i)
#define GSUB_T0()
#define GSUB(c, i)
#define GCOUNT(c, i, n)
#endif

// cold per-chain state kept in LDS (every lane of the group reads the same word: a broadcast)
__device__ __forceinline__ Stream ld_stream(const SpecLds &S, int gi) {
  LDSP(uint32_t) w = S.gstream + gi * 4;
  Stream st;
  st.k0 = w[0];
  st.k1 = w[1];
  st.c2 = w[2];
  st.c3 = w[3];
  return st;
}
#define C_INB(S, gi) ((S).gval[(gi) * GV_N + GV_INB])
#define C_AMASK(c) ((1u << (c).bits) - 1u)
#define C_KEYBITS(c) ((c).bits * (c).Mh)

template <int KT>
__device__ __forceinline__ void genotype_changed(Grp<KT> &c) {
  c.mvalid = false;
  c.gen += 1;  // the interval memo of the group is wiped by memo_wipe() before it is read again
}

// ---- decision contexts per genotype (round 5) ----
// A chain that never settles (phase-ambiguous samples, shallow pileups) keeps coming back to genotypes it has held before: at
// docs/example's slowest units 98 % of the accepted moves lead to one of the last 64 ORDERED genotypes, at the shallow synthetic
// units (40 reads of quality 3-20) 95 % to one of the last 32 (tests/aids/analyze_moves.py: the oracle's sub-step log).  Everything a
// sub-step decides with is a pure function of the ordered genotype g it starts from (mutation.py:60-161): for sub-step e = (h, j)
// the move probability pr(g, e) and the likelihood of the genotype the move leads to; for an interval step (structural.py:490-587)
// the total move probability tot(g, type, start, stop).  Without contexts every accepted move starts a new speculation round --
// proposals, cache probes, the evaluations of the misses -- only to re-derive these numbers, and wipes the interval memo.
// With them (CTX instantiations: the resumed chains of the phased sampler, one chain per wavefront, biallelic units) the chain
// keeps up to SPEC_CTX_MAX contexts in its region of the workspace -- slot = {K words, pr[K M], llk[K M], tot[2][M (M + 1) / 2]} --
// and the current genotype's in LDS (CXP_PR / CXP_LLK, memo_tot).  A move into a genotype with a context is a context switch: one
// directory look-up (the tags of the slots, one per lane, a ballot) and one round trip that loads pr / llk; the round after it is
// a ballot over stored numbers.  What is not known yet (-1 / NaN) is evaluated as before and written through.  Results-neutral
// by the argument of the likelihood cache: a stored value is what this arithmetic computed for that ordered genotype (tuning
// flag 524288 switches the contexts off: same traces -- tests/test_gpu_moving_chains.py).
constexpr int SPEC_CTX_MAX = 64;   // contexts per chain (one directory entry per lane)
constexpr int SPEC_CTX_HDR = 8;    // 8-byte words at the head of a slot: the genotype's haplotype words (K <= 8)
#ifndef MCHAP_CTX_GEN
#define MCHAP_CTX_GEN 24           // genotype changes (in this launch) after which a chain keeps contexts
#endif
// the chain's context state (cold, kept out of the registers): [CX_CUR] the slot whose context the LDS holds (+ 1; 0: none),
// [CX_GEN] the generation it was looked up for, [CX_CLOCK] the LRU clock, and the chain's trial -- [CX_HITS] contexts found among
// the [CX_LOOKS] look-ups of the current window, [CX_PAUSE] the generation until which the chain does without contexts (a chain
// that wanders -- shallow units visit thousands of genotypes once -- pays for look-ups that miss)
enum { CX_CUR = 0, CX_GEN, CX_CLOCK, CX_HITS, CX_LOOKS, CX_PAUSE, CX_FAILS, CX_N = 8 };
#define CXP_TAG(S) ((LDSP(uint64_t))(S).cx_lds)
#define CXP_STAMP(S) ((LDSP(uint32_t))((S).cx_lds + 8 * SPEC_CTX_MAX))
#define CXP_ST(S) ((LDSP(uint32_t))((S).cx_lds + 12 * SPEC_CTX_MAX))
#define CXP_PR(S) ((S).cx_val)
#define CXP_LLK(S, nmax) (CXP_PR(S) + (nmax))
constexpr uint32_t SPEC_CTX_TRIAL = 64;     // look-ups per trial window
constexpr uint32_t SPEC_CTX_MIN_HITS = 24;  // ... of which at least this many must find their context, or the chain pauses
constexpr uint32_t SPEC_CTX_PAUSE = 2048;   // ... for this many genotype changes
__host__ __device__ inline int spec_ctx_words(int K, int Mmax) { return SPEC_CTX_HDR + 2 * K * Mmax + 2 * spec_memo_entries(Mmax); }
__host__ __device__ inline size_t spec_ctx_lds_bytes(int K, int Mmax) { return (size_t)SPEC_CTX_MAX * 12 + (size_t)16 * K * Mmax + 4 * CX_N + 16; }
// The contexts are written and read by different lanes of the same wavefront at different times: both sides go to the L2
// (agent scope), and an lds_sync() -- which waits for the wave's outstanding stores -- separates a context's writes from its loads.
__device__ __forceinline__ uint64_t ctx_ld(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctx_st(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctx_st(uint64_t *p, double v) { ctx_st(p, (uint64_t)__double_as_longlong(v)); }

// Make the context in LDS (CXP_PR / CXP_LLK) describe the chain's current genotype (every lane of the wave calls; one chain per wave).
// Returns true if the context is new (nothing known: the caller's interval memo starts empty as well).
template <int KT>
__device__ __forceinline__ bool ctx_acquire(Grp<KT> &c, const SpecLds &S, int n, int mmax, int lane) {
  const int cx_nmax = KT * mmax, cx_words = spec_ctx_words(KT, mmax);
  if (CXP_ST(S)[CX_GEN] == c.gen && CXP_ST(S)[CX_CUR] != 0u) return false;
  STAT_WAVE(3, 1);
  GT0(t_acq);
  lds_sync();  // (the wave's write-through stores have reached the L2; nobody reads the context in LDS any more)
  const GWords<KT> g = c.g;
  const bool wide = C_KEYBITS(c) * KT > 63;
  const uint64_t tag = tag_of<KT>(g, C_KEYBITS(c));
  const uint64_t mine = lane < S.cx_n ? CXP_TAG(S)[lane] : 0ull;
  const uint32_t stamp = lane < S.cx_n ? CXP_STAMP(S)[lane] : 0xFFFFFFFFu;
  unsigned long long hit = __ballot(mine == tag);
  int slot = hit ? __ffsll((long long)hit) - 1 : -1;
  bool fresh = false;
  uint64_t *sp = nullptr;
  if (slot >= 0) {
    sp = S.cx_base + (size_t)slot * cx_words;
    uint64_t vp[3], vl[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {  // (n <= 192)
      const int e = lane + WAVE * t;
      vp[t] = e < n ? ctx_ld(sp + SPEC_CTX_HDR + e) : 0ull;
      vl[t] = e < n ? ctx_ld(sp + SPEC_CTX_HDR + cx_nmax + e) : 0ull;
    }
    bool same = true;
    if (wide) {  // the tag is a hash: the slot's words decide
      const uint64_t w = lane < KT ? ctx_ld(sp + lane) : 0ull;
      same = __ballot(lane < KT && w != sel_word<KT>(g, lane)) == 0ull;
    }
    if (same) {
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int e = lane + WAVE * t;
        if (e < n) {
          CXP_PR(S)[e] = __longlong_as_double((long long)vp[t]);
          CXP_LLK(S, cx_nmax)[e] = __longlong_as_double((long long)vl[t]);
        }
      }
    } else {
      fresh = true;  // another genotype with this tag: its slot is taken over
    }
  } else {
    // least recently used slot (an empty one first: stamp 0); ties go to the lowest lane
    uint32_t key = stamp, best = stamp;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, WAVE));
    const unsigned long long who = __ballot(key == best && lane < S.cx_n);
    slot = __ffsll((long long)who) - 1;
    sp = S.cx_base + (size_t)slot * cx_words;
    fresh = true;
  }
  const uint32_t clock = CXP_ST(S)[CX_CLOCK] + 1u;
  uint32_t looks = CXP_ST(S)[CX_LOOKS] + 1u, hits = CXP_ST(S)[CX_HITS] + (fresh ? 0u : 1u), pause = CXP_ST(S)[CX_PAUSE];
  uint32_t fails = CXP_ST(S)[CX_FAILS];
  if (looks >= SPEC_CTX_TRIAL) {  // (the slots stay as they are: a context never goes stale)
    if (hits < SPEC_CTX_MIN_HITS) {  // each failed trial doubles the pause
      pause = c.gen + (SPEC_CTX_PAUSE << (fails < 12u ? fails : 12u));
      fails += 1u;
    }
    looks = 0;
    hits = 0;
  }
  if (lane == slot) {
    CXP_TAG(S)[lane] = tag;
    CXP_STAMP(S)[lane] = clock;
  }
  if (fresh) {
    if (lane < KT) ctx_st(sp + lane, sel_word<KT>(g, lane));
    for (int e = lane; e < n; e += WAVE) {
      ctx_st(sp + SPEC_CTX_HDR + e, -1.0);
      CXP_PR(S)[e] = -1.0;
    }
    const int ms = cx_words - SPEC_CTX_HDR - 2 * cx_nmax;
    for (int i = lane; i < ms; i += WAVE) ctx_st(sp + SPEC_CTX_HDR + 2 * cx_nmax + i, (double)NAN);
  }
  lds_sync();  // (every lane has read the state)
  if (lane == 0) {
    CXP_ST(S)[CX_CUR] = (uint32_t)slot + 1u;
    CXP_ST(S)[CX_GEN] = c.gen;
    CXP_ST(S)[CX_CLOCK] = clock;
    CXP_ST(S)[CX_LOOKS] = looks;
    CXP_ST(S)[CX_HITS] = hits;
    CXP_ST(S)[CX_PAUSE] = pause;
    CXP_ST(S)[CX_FAILS] = fails;
  }
  lds_sync();
  STAT_WAVE(4, fresh ? 1 : 0);
  GT1(c, 13, t_acq);
  return fresh;
}

template <int KT, int RPL>
__device__ __forceinline__ double spec_coop_body(const SpecLds &S, int src, int sg, int mmax, int Mh, uint32_t amask,
                                                 GLBP(const double) rt, GLBP(const double) cw, int rpad, int lane,
                                                 int nrd, bool grouped) {
  // nrd: reads left from this block's first read on.  With -DMCHAP_MASK_PADDING lanes whose read is padding keep the
  // neutral 1.0 / count 0 without loading the padded tail of the row: 33 % fewer HBM bytes at config #2 (R = 200 in
  // rows of 256) but 2.5 % slower (a compare and an exec update per load), so it is off by default.
  // rt / cw already point at the lane's first read of the block of RPL chunks
  constexpr int UNR = MCHAP_COOP_UNR;
  const int n_pairs = KT * Mh;
  const double invK = 1.0 / (double)KT;
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
        const uint64_t wh = S.pw[(size_t)h * WAVE + src];
        const uint32_t a = (uint32_t)(wh >> S.shift[(size_t)sg * mmax + j]) & amask;
        myrow = (int)S.cols[(size_t)sg * mmax + j] + (int)a;
      }
    }
    const int lim = min(WAVE, n_pairs - base);
    int jj = base % Mh;
    for (int q0 = 0; q0 < lim; q0 += UNR) {
      double v[UNR][RPL];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int q = q0 + u;
        const int row = __builtin_amdgcn_readlane(myrow, q < lim ? q : 0);
        GLBP(const double) rp = rt + (size_t)row * rpad;
#pragma unroll
        for (int i = 0; i < RPL; i++) v[u][i] = MCHAP_PAD_LANE(lane + WAVE * i < nrd) ? rp[WAVE * i] : 1.0;
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        if (q0 + u < lim) {
#pragma unroll
          for (int i = 0; i < RPL; i++) prod[i] *= v[u][i];
          if (++jj == Mh) {
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
  for (int i = 0; i < RPL; i++) wv[i] = MCHAP_PAD_LANE(lane + WAVE * i < nrd) ? cw[WAVE * i] : 0.0;
  return read_log_sum<RPL>(acc, wv, grouped);  // per-lane partial sum; the caller reduces across the wave
}

// The same evaluation from the coded table: one load per (haplotype, position) pair fetches the codes of the
// lane's RPL reads (RPL bytes, lane-major layout), the float64 factors come from the unit's dictionary in LDS.
// Same factors in the same order, hence the same value as spec_coop_body.
template <int KT, int RPL, class CT, bool LT = false>
__device__ __forceinline__ double spec_coop_coded(const SpecLds &S, int src, int sg, int mmax, int Mh, uint32_t amask,
                                                  typename TabPtr<LT>::u8 ct, typename TabPtr<LT>::f64 cw, int crow, int lane, bool grouped) {
  // ct points at the lane's first code of the block of RPL chunks; cw at the lane's first read of the block;
  // crow = bytes per row of the coded table
  constexpr int UNR = MCHAP_CODED_UNR;
  LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
  const int n_pairs = KT * Mh;
  const double invK = 1.0 / (double)KT;
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
        const uint64_t wh = S.pw[(size_t)h * WAVE + src];
        const uint32_t a = (uint32_t)(wh >> S.shift[(size_t)sg * mmax + j]) & amask;
        myrow = (int)S.cols[(size_t)sg * mmax + j] + (int)a;
      }
    }
    const int lim = min(WAVE, n_pairs - base);
    int jj = base % Mh;
    for (int q0 = 0; q0 < lim; q0 += UNR) {
      CT cd[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int q = q0 + u;
        const int row = __builtin_amdgcn_readlane(myrow, q < lim ? q : 0);
        cd[u] = TabPtr<LT>::template ld<CT>(ct + (size_t)row * crow);
      }
      // the dictionary gathers of CG pairs are issued together, ahead of the (scalar) branches on the haplotype
      // boundaries: one LDS round trip per CG pairs instead of one per pair (as in spec_hap_prod)
      constexpr int CG = KT <= 4 ? 2 : MCHAP_SPEC_CG;  // measured: 4 costs ploidy <= 4 more in spills than it hides
#pragma unroll
      for (int u0 = 0; u0 < UNR; u0 += CG) {
        if (q0 + u0 < lim) {
          double f[CG][RPL];
#pragma unroll
          for (int u = 0; u < CG; u++)
#pragma unroll
            for (int i = 0; i < RPL; i++) f[u][i] = dict[((uint32_t)cd[u0 + u] >> (8 * i)) & 255u];
#pragma unroll
          for (int u = 0; u < CG; u++) {
            if (q0 + u0 + u < lim) {
#pragma unroll
              for (int i = 0; i < RPL; i++) prod[i] *= f[u][i];
              if (++jj == Mh) {
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
    }
  }
  double wv[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) wv[i] = cw[WAVE * i];
  return read_log_sum<RPL>(acc, wv, grouped);
}

// ---- coded evaluation, one haplotype at a time ----
// Row of pair p = h * Mh + j of the genotype whose words are words[h] (an LDS array), for the lane's pairs
// p = lane, lane + 64 and lane + 128 (K * Mh <= 192).
struct PairRows {
  int r0, r1, r2;
};
template <int KT>
__device__ __forceinline__ PairRows spec_pair_rows(LDSP(const uint64_t) words, int wstride, int widx, const SpecLds &S, int sg, int mmax,
                                                   int Mh, uint32_t amask, int lane) {
  const int n_pairs = KT * Mh;
  PairRows R;
  R.r0 = 0;
  R.r1 = 0;
  R.r2 = 0;
#pragma unroll
  for (int t = 0; t < (MCHAP_SPEC_DEEP ? 3 : 2); t++) {
    const int p = lane + WAVE * t;
    if (p < n_pairs) {  // (t = 2, the "deep" instantiation: octoploids with more than 16 sampled positions)
      const int h = p / Mh, j = p - h * Mh;
      const uint64_t wh = words[(size_t)h * wstride + widx];
      const uint32_t a = (uint32_t)(wh >> S.shift[(size_t)sg * mmax + j]) & amask;
      const int r = (int)S.cols[(size_t)sg * mmax + j] + (int)a;
      if (t == 0) R.r0 = r;
      else if (t == 1) R.r1 = r;
      else R.r2 = r;
    }
  }
  return R;
}
// prod[i] = product over the Mh positions of haplotype h (pairs h*Mh .. h*Mh+Mh-1, rows in `rows`), reads of
// chunk i, in position order: the factors and their order are those of spec_coop_coded
template <int RPL, class CT, bool LT = false>
__device__ __forceinline__ void spec_hap_prod(LDSP(double) dict, const PairRows rows, int p0, int Mh, typename TabPtr<LT>::u8 ct,
                                              int crow, double (&prod)[RPL]) {
  constexpr int UNR = 8;  // code loads in flight
  constexpr int GB = 2;   // positions whose dictionary gathers are in flight together
#pragma unroll
  for (int i = 0; i < RPL; i++) prod[i] = 1.0;
  for (int j0 = 0; j0 < Mh; j0 += UNR) {
    CT cd[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int p = __builtin_amdgcn_readfirstlane(p0 + min(j0 + u, Mh - 1));
      const int row = p < WAVE ? __builtin_amdgcn_readlane(rows.r0, p & (WAVE - 1))
                               : (p < 2 * WAVE ? __builtin_amdgcn_readlane(rows.r1, p & (WAVE - 1)) : __builtin_amdgcn_readlane(rows.r2, p & (WAVE - 1)));
      cd[u] = TabPtr<LT>::template ld<CT>(ct + (size_t)row * crow);
    }
    // No branch between the positions: the gathers of GB positions are issued together (a branch per position made
    // every gather a full LDS round trip: 2/3 of an evaluation's time) and a position beyond Mh multiplies by 1.0
    // (exact), so the products are those of the plain loop, factor for factor.
#pragma unroll
    for (int u0 = 0; u0 < UNR; u0 += GB) {
      if (j0 + u0 < Mh) {
        double f[GB][RPL];
#pragma unroll
        for (int u = 0; u < GB; u++)
#pragma unroll
          for (int i = 0; i < RPL; i++) f[u][i] = dict[((uint32_t)cd[u0 + u] >> (8 * i)) & 255u];
#pragma unroll
        for (int u = 0; u < GB; u++) {
          const bool on = j0 + u0 + u < Mh;
#pragma unroll
          for (int i = 0; i < RPL; i++) prod[i] *= on ? f[u][i] : 1.0;
        }
      }
    }
  }
}
// Haplotype products of a chain's current genotype (first block of up to 4 read chunks): K x 4 doubles per lane, in
// registers -- or, with one chain per wave (G = 64), in the wave's LDS cache, each lane its own column: K x 4 doubles
// are 64 VGPRs at K = 4 and 128 at K = 8, which the evaluation code does not have to spare.
template <int KT, bool IN_LDS>
struct BaseProducts;
template <int KT>
struct BaseProducts<KT, false> {
  double v[KT][4];
  __device__ __forceinline__ double get(int h, int i, int) const { return v[h][i]; }
  __device__ __forceinline__ void set(int h, int i, int, double x) { v[h][i] = x; }
};
template <int KT>
struct BaseProducts<KT, true> {
  LDSP(double) p;
  __device__ __forceinline__ double get(int h, int i, int lane) const { return p[(h * 4 + i) * WAVE + lane]; }
  __device__ __forceinline__ void set(int h, int i, int lane, double x) { p[(h * 4 + i) * WAVE + lane] = x; }
};
// ... and for the read chunks beyond the first four (deep units: 257 reads and more), in the chain's rows of the
// workspace (SimtParams::gbp, [chain][K][rpad]): each lane its own column again, so a lane only ever reads what it wrote.
template <int KT>
struct BaseProductsG {
  GLBP(double) p;  // the chain's [K][rpad] rows
  int rpad, cb;    // first chunk of the block the products belong to
  __device__ __forceinline__ double get(int h, int i, int lane) const { return p[(size_t)h * rpad + (cb + i) * WAVE + lane]; }
  __device__ __forceinline__ void set(int h, int i, int lane, double x) { p[(size_t)h * rpad + (cb + i) * WAVE + lane] = x; }
};
// One request with reuse: haplotypes whose word equals the base word take the base product bp[h].
template <int KT, int RPL, class CT, bool LT = false, class BP>
__device__ __forceinline__ double spec_coop_reuse(const SpecLds &S, int src, int sg, int mmax, int Mh, uint32_t amask,
                                                  typename TabPtr<LT>::u8 ct, typename TabPtr<LT>::f64 cw, int crow, int lane,
                                                  const BP &bp, bool use_base, bool grouped) {
  LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
  const double invK = 1.0 / (double)KT;
  const PairRows rows = spec_pair_rows<KT>(S.pw, WAVE, src, S, sg, mmax, Mh, amask, lane);
  double acc[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) acc[i] = 0.0;
#pragma unroll
  for (int h = 0; h < KT; h++) {
    const uint64_t a = S.pw[(size_t)h * WAVE + src], b = S.bw[(size_t)sg * KT + h];
    const bool same = use_base && __builtin_amdgcn_readfirstlane((int)(a == b)) != 0;
    double ph[RPL];
    if (same) {
#pragma unroll
      for (int i = 0; i < RPL; i++) ph[i] = bp.get(h, i, lane);
    } else {
      spec_hap_prod<RPL, CT, LT>(dict, rows, h * Mh, Mh, ct, crow, ph);
    }
#pragma unroll
    for (int i = 0; i < RPL; i++) acc[i] += ph[i] * invK;
  }
  double wv[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) wv[i] = cw[WAVE * i];
  return read_log_sum<RPL>(acc, wv, grouped);
}
// The same for a block of deep chunks, whose base products come from the workspace: the K x RPL loads of a block go out
// together, ahead of everything else (behind the per-haplotype branches each would be a memory round trip of its own:
// measured 54 000 ticks per evaluation at config #5, of which 32 dependent round trips).  A changed haplotype's row is loaded
// too and not used.
template <int KT, int RPL, class CT, bool LT = false>
__device__ __forceinline__ double spec_coop_reuse_g(const SpecLds &S, int src, int sg, int mmax, int Mh, uint32_t amask,
                                                    typename TabPtr<LT>::u8 ct, typename TabPtr<LT>::f64 cw, int crow, int lane,
                                                    const BaseProductsG<KT> &bg, bool grouped) {
  double bpv[KT][RPL];
#pragma unroll
  for (int h = 0; h < KT; h++)
#pragma unroll
    for (int i = 0; i < RPL; i++) bpv[h][i] = bg.get(h, i, lane);
  LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
  const double invK = 1.0 / (double)KT;
  const PairRows rows = spec_pair_rows<KT>(S.pw, WAVE, src, S, sg, mmax, Mh, amask, lane);
  double acc[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) acc[i] = 0.0;
#pragma unroll
  for (int h = 0; h < KT; h++) {
    const uint64_t a = S.pw[(size_t)h * WAVE + src], b = S.bw[(size_t)sg * KT + h];
    const bool same = __builtin_amdgcn_readfirstlane((int)(a == b)) != 0;
    double ph[RPL];
    if (same) {
#pragma unroll
      for (int i = 0; i < RPL; i++) ph[i] = bpv[h][i];
    } else {
      spec_hap_prod<RPL, CT, LT>(dict, rows, h * Mh, Mh, ct, crow, ph);
    }
#pragma unroll
    for (int i = 0; i < RPL; i++) acc[i] += ph[i] * invK;
  }
  double wv[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) wv[i] = cw[WAVE * i];
  return read_log_sum<RPL>(acc, wv, grouped);
}
template <int KT, int RPL, class CT, bool LT = false, class BP>
__device__ __forceinline__ void spec_base_products(const SpecLds &S, int sg, int mmax, int Mh, uint32_t amask,
                                                   typename TabPtr<LT>::u8 ct, int crow, int lane, BP &bp) {
  LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
  const PairRows rows = spec_pair_rows<KT>(S.bw + (size_t)sg * KT, 1, 0, S, sg, mmax, Mh, amask, lane);
#pragma unroll
  for (int h = 0; h < KT; h++) {
    double ph[RPL];
    spec_hap_prod<RPL, CT, LT>(dict, rows, h * Mh, Mh, ct, crow, ph);
#pragma unroll
    for (int i = 0; i < RPL; i++) bp.set(h, i, lane, ph[i]);
  }
}
// The same for ONE haplotype h of the current genotype (the rows of the deep chunks are refreshed haplotype by haplotype:
// a move changes one or two words)
template <int KT, int RPL, class CT, bool LT = false, class BP>
__device__ __forceinline__ void spec_base_products_of(const SpecLds &S, const PairRows rows, int sg, int h, int Mh,
                                                      typename TabPtr<LT>::u8 ct, int crow, int lane, BP &bp) {
  LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
  double ph[RPL];
  spec_hap_prod<RPL, CT, LT>(dict, rows, h * Mh, Mh, ct, crow, ph);
#pragma unroll
  for (int i = 0; i < RPL; i++) bp.set(h, i, lane, ph[i]);
}

// lanes of `reqs` whose request words equal those of lane `src` (src included)
template <int KT>
__device__ __forceinline__ unsigned long long spec_same_request(LDSP(uint64_t) pwbuf, unsigned long long reqs, int src, int lane) {
  if ((reqs & (reqs - 1ull)) == 0ull) return reqs;  // a single request
  // (no short-circuit: the 2 K LDS reads go out together instead of one round trip per word behind a branch)
  bool eq = ((reqs >> lane) & 1ull) != 0ull;
#pragma unroll
  for (int h = 0; h < KT; h++) eq = eq & (pwbuf[(size_t)h * WAVE + lane] == pwbuf[(size_t)h * WAVE + src]);
  return __ballot(eq);
}

// Serves every request of the wave (bit mask `todo`), chain by chain, with all 64 lanes.  Inlined on purpose: see the
// note on device function calls in DESIGN.md section 4.1.
// DEDUP = false: the caller's requests are distinct by construction (the proposals of a mutation round: different sub-steps of
// one genotype) -- the comparisons that find equal requests are left out
template <int KT, int G, bool LT = false, bool DEDUP = true>
__device__ __forceinline__ double spec_coop_all(unsigned long long todo, LDSP(uint64_t) pwbuf, LDSP(uint8_t) shift_tab,
                                                LDSP(uint16_t) cols_tab, LDSP(uint16_t) nreads_tab, LDSP(uint16_t) ndict_tab,
                                                LDSP(double) dict_tab, LDSP(uint64_t) gptr_tab, LDSP(uint64_t) bw_tab,
                                                bool reuse, int crow, int mmax, int Mh_lane, uint32_t amask_lane, int rpad,
                                                int lane, LDSP(const uint8_t) lds_ct = nullptr, LDSP(const double) lds_cw = nullptr,
                                                LDSP(double) bpc = nullptr, LDSP(uint64_t) bpt = nullptr,
                                                LDSP(const uint8_t) sct = nullptr, LDSP(uint64_t) gbt = nullptr) {
  SpecLds S;
  S.pw = pwbuf;
  S.shift = shift_tab;
  S.cols = cols_tab;
  S.dict = dict_tab;
  S.bw = bw_tab;
  const int nch_batch = rpad / WAVE;
  double val = 0.0;
  // Reuse: a chain's proposals differ from its current genotype (the base words in bw_tab) in one or two haplotype
  // words; when a chain has several requests in this call, the products of the base haplotypes are formed once and a
  // request only forms the products of the words it changed.  Same factors, same order: bit-identical values.
  while (todo) {
    // all requests of one chain (the lanes of a group are contiguous)
    const int sg = (__ffsll((long long)todo) - 1) / G;
    const unsigned long long gmask = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << (sg * (G & 63)));
    unsigned long long reqs = todo & gmask;
    todo &= ~gmask;
    const int first = __ffsll((long long)reqs) - 1;
    const int Mh = __builtin_amdgcn_readfirstlane(__shfl(Mh_lane, first, WAVE));
    const uint32_t amask = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)amask_lane, first, WAVE));
    LDSP(uint64_t) gp = gptr_tab + sg * GP_N;  // the requesting chain's pointers (wave-uniform)
    GLBP(const double) rt = (GLBP(const double))(uintptr_t)gp[GP_RT] + lane;
    typename TabPtr<LT>::f64 cw;
    if constexpr (LT) cw = lds_cw + lane;
    else cw = (GLBP(const double))(uintptr_t)gp[GP_CW] + lane;
    const int nrd = (int)nreads_tab[sg];
    const bool grouped = nd_w01(ndict_tab[sg]);  // the unit's weights are 0 / 1: one logarithm per lane and block of chunks
    // read chunks of THIS unit: a batch is padded to its deepest unit, but the chunks beyond a unit's own reads hold
    // padding only -- weight 0, terms +-0.0, which every sum absorbs exactly -- so they are not evaluated (a ragged
    // batch of real pileups is mostly shallow units: docs/example has 2 to 534 read pairs per unit)
    const int nch = max(1, min(nch_batch, (nrd + WAVE - 1) / WAVE));
    const int cstride = crow / WAVE;  // code bytes per lane and row
    if (MCHAP_REUSE_MAXK >= KT && nd_count(ndict_tab[sg]) != 0 && KT * Mh <= (MCHAP_SPEC_DEEP ? 3 : 2) * WAVE) {
      // coded table, one haplotype at a time; with use_base the haplotypes a request did not change are skipped.
      // The base products cover the first block of (up to) 4 chunks; deeper reads add their other blocks in full.
      const int nb0 = nch < 4 ? nch : 4;
      typename TabPtr<LT>::u8 ct;
      if constexpr (LT) ct = lds_ct + (size_t)lane * cstride;
      else ct = (GLBP(const uint8_t))(uintptr_t)gp[GP_CT] + (size_t)lane * cstride;
      // One chain per wave (G = 64): the base products live in the wave's LDS cache and stay there between calls, so
      // the rounds of a compound step that does not move, and of a fill, form them once.  Same values either way.
      constexpr bool BPL = (G == 64) && !LT;
      bool cached = false;
      if (BPL && bpc != nullptr) {
        bool eq = bpt[KT] != 0ull;
#pragma unroll
        for (int h = 0; h < KT; h++) eq = eq & (bpt[h] == bw_tab[(size_t)sg * KT + h]);
        cached = __builtin_amdgcn_readfirstlane((int)eq) != 0;
      }
      const bool use_base = reuse && (!BPL || bpc != nullptr) && (cached || __popcll(reqs) >= 2);
      BaseProducts<KT, BPL> bp;
      if constexpr (BPL) {
        bp.p = bpc;
      } else {
#pragma unroll
        for (int h = 0; h < KT; h++)
#pragma unroll
          for (int i = 0; i < 4; i++) bp.v[h][i] = 0.0;
      }
      if (use_base && !cached) {
        if constexpr (BPL) {
          // Only the haplotypes whose word changed since their products were formed: an accepted mutation changes one word,
          // and in a chain that keeps moving nearly every round follows a move -- re-forming all K products each time was
          // half of such a round (phase timers, 16-read units: 12 000 ticks per round, 6 000 of them here).
          const bool any_valid = bpt[KT] != 0ull;
#if MCHAP_SPEC_SBS
          if (sct != nullptr && nch == 1) {
            // the unit's codes are in LDS (at most 64 reads, lane = read): the changed haplotypes' products without the memory
            // round trip of the code loads -- after every accepted move of a chain that keeps moving.  The factors and their
            // order are spec_hap_prod's (a position beyond Mh multiplies by 1.0: exact), hence the same products.
            LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
            LDSP(const uint8_t) shift = shift_tab + (size_t)sg * mmax;
            LDSP(const uint16_t) cols = cols_tab + (size_t)sg * mmax;
            for (int h = 0; h < KT; h++) {
              const uint64_t wb = bw_tab[(size_t)sg * KT + h];
              const bool ok = any_valid && bpt[h] == wb;
              if (__builtin_amdgcn_readfirstlane((int)ok) != 0) continue;
              double ph = 1.0;
              for (int j0 = 0; j0 < Mh; j0 += 8) {
                uint8_t cd[8];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                  const int j = min(j0 + t, Mh - 1);
                  cd[t] = sct[sct_off((uint32_t)cols[j] + ((uint32_t)(wb >> shift[j]) & amask)) + lane];
                }
                double f[8];
#pragma unroll
                for (int t = 0; t < 8; t++) f[t] = dict[cd[t]];
#pragma unroll
                for (int t = 0; t < 8; t++) ph *= (j0 + t < Mh) ? f[t] : 1.0;
              }
              bp.set(h, 0, lane, ph);
            }
          } else
#endif
          {
          const PairRows rows = spec_pair_rows<KT>(bw_tab + (size_t)sg * KT, 1, 0, S, sg, mmax, Mh, amask, lane);
          for (int h = 0; h < KT; h++) {
            const bool ok = any_valid && bpt[h] == bw_tab[(size_t)sg * KT + h];
            if (__builtin_amdgcn_readfirstlane((int)ok) != 0) continue;
            if (nb0 == 1) spec_base_products_of<KT, 1, uint8_t, LT>(S, rows, sg, h, Mh, ct, crow, lane, bp);
            else if (nb0 == 2) spec_base_products_of<KT, 2, uint16_t, LT>(S, rows, sg, h, Mh, ct, crow, lane, bp);
            else if (nb0 == 3) spec_base_products_of<KT, 3, uint32_t, LT>(S, rows, sg, h, Mh, ct, crow, lane, bp);
            else spec_base_products_of<KT, 4, uint32_t, LT>(S, rows, sg, h, Mh, ct, crow, lane, bp);
          }
          }
          lds_sync();
          if (lane == 0) {
#pragma unroll
            for (int h = 0; h < KT; h++) bpt[h] = bw_tab[(size_t)sg * KT + h];
            bpt[KT] = 1ull;
          }
          lds_sync();
        } else {
          if (nb0 == 1) spec_base_products<KT, 1, uint8_t, LT>(S, sg, mmax, Mh, amask, ct, crow, lane, bp);
          else if (nb0 == 2) spec_base_products<KT, 2, uint16_t, LT>(S, sg, mmax, Mh, amask, ct, crow, lane, bp);
          else if (nb0 == 3) spec_base_products<KT, 3, uint32_t, LT>(S, sg, mmax, Mh, amask, ct, crow, lane, bp);
          else spec_base_products<KT, 4, uint32_t, LT>(S, sg, mmax, Mh, amask, ct, crow, lane, bp);
        }
      }
      // Deep units: the products of the current genotype's haplotypes for the chunks beyond the first four are kept in the
      // chain's rows of the workspace, refreshed haplotype by haplotype when a word differs from the one they were formed
      // for -- a request then forms the products of the words it changed only, in every chunk, instead of all K x Mh
      // factors of every read of twelve chunks (config #5: 2 560 gathers per lane and request -> 320 + 112 loads).
      GLBP(double) gbp = (GLBP(double))(uintptr_t)gp[GP_GBP];
      const bool deep = MCHAP_SPEC_DEEP && use_base && nch > 4 && gbp != nullptr && gbt != nullptr;
      if (deep) {
        LDSP(uint64_t) gt = gbt + (size_t)sg * (KT + 1);
        const bool any_valid = gt[KT] != 0ull;
        bool have_rows = false, wrote = false;
        PairRows rows;
        rows.r0 = rows.r1 = rows.r2 = 0;
        for (int h = 0; h < KT; h++) {
          const bool ok = any_valid && gt[h] == bw_tab[(size_t)sg * KT + h];
          if (__builtin_amdgcn_readfirstlane((int)ok) != 0) continue;
          if (!have_rows) {
            rows = spec_pair_rows<KT>(bw_tab + (size_t)sg * KT, 1, 0, S, sg, mmax, Mh, amask, lane);
            have_rows = true;
          }
          for (int cb = 4; cb < nch; cb += 4) {
            const int rem = nch - cb;
            BaseProductsG<KT> bg;
            bg.p = gbp;
            bg.rpad = rpad;
            bg.cb = cb;
            if (rem >= 4) spec_base_products_of<KT, 4, uint32_t, LT>(S, rows, sg, h, Mh, ct + cb, crow, lane, bg);
            else if (rem == 3) spec_base_products_of<KT, 3, uint32_t, LT>(S, rows, sg, h, Mh, ct + cb, crow, lane, bg);
            else if (rem == 2) spec_base_products_of<KT, 2, uint16_t, LT>(S, rows, sg, h, Mh, ct + cb, crow, lane, bg);
            else spec_base_products_of<KT, 1, uint8_t, LT>(S, rows, sg, h, Mh, ct + cb, crow, lane, bg);
          }
          wrote = true;
        }
        if (wrote) {
          lds_sync();
          if (lane == 0) {
#pragma unroll
            for (int h = 0; h < KT; h++) gt[h] = bw_tab[(size_t)sg * KT + h];
            gt[KT] = 1ull;
          }
          lds_sync();
        }
      }
      // Shallow units (one chunk of reads, at most 32 of them): lanes over reads leave most of the wavefront idle, so
      // two (17-32 reads) or four (up to 16) requests are evaluated side by side, a sub-group of 32 / 16 lanes each.
      // A lane forms the same factors in the same order for its (request, read); the sum over the reads is the
      // wavefront butterfly restricted to the sub-group -- the other lanes of a full-width evaluation hold padding
      // reads with weight 0, whose terms are +0.0 -- so the values are those of the one-request path, bit for bit.
#if MCHAP_SPEC_SBS
      if constexpr (BPL) {
        if (use_base && nch == 1 && nrd <= 32) {
          const int RS = nrd <= 16 ? 16 : 32, NQ = WAVE / RS;
          const int sub = lane / RS, r = lane % RS;
          LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
          LDSP(const uint8_t) shift = shift_tab + (size_t)sg * mmax;
          LDSP(const uint16_t) cols = cols_tab + (size_t)sg * mmax;
          GLBP(const uint8_t) ctb = (GLBP(const uint8_t))(uintptr_t)gp[GP_CT];
          const double cwr = ((GLBP(const double))(uintptr_t)gp[GP_CW])[r];
          const double invK = 1.0 / (double)KT;
          while (reqs) {
            int srcs[4] = {0, 0, 0, 0};
            unsigned long long dupm[4] = {0ull, 0ull, 0ull, 0ull};
            int nq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (k < NQ && reqs) {
                const int src = __ffsll((long long)reqs) - 1;
                const unsigned long long dups = (DEDUP ? spec_same_request<KT>(pwbuf, reqs, src, lane) : (1ull << src));
                reqs &= ~dups;
                srcs[k] = src;
                dupm[k] = dups;
                nq = k + 1;
              }
            }
            const int my_src = sub == 0 ? srcs[0] : (sub == 1 ? srcs[1] : (sub == 2 ? srcs[2] : srcs[3]));
            // the product over the positions of one haplotype word for the lane's read, eight positions at a time: their
            // code loads go out together, then the dictionary gathers, then the products in position order (a position
            // beyond Mh multiplies by 1.0: exact) -- one memory round trip per eight positions instead of one per position
            auto hap_product = [&](uint64_t wa) -> double {
              double ph = 1.0;
              for (int j0 = 0; j0 < Mh; j0 += 8) {
                uint8_t cd[8];
                uint32_t row[8];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                  const int j = min(j0 + t, Mh - 1);
                  row[t] = (uint32_t)cols[j] + ((uint32_t)(wa >> shift[j]) & amask);
                }
                if (sct != nullptr) {  // (wave-uniform) the unit's codes are in LDS: no memory round trip at all
#pragma unroll
                  for (int t = 0; t < 8; t++) cd[t] = sct[sct_off(row[t]) + r];
                } else {
#pragma unroll
                  for (int t = 0; t < 8; t++) cd[t] = ctb[(size_t)(row[t] * WAVE + r) * cstride];
                }
                double f[8];
#pragma unroll
                for (int t = 0; t < 8; t++) f[t] = dict[cd[t]];
#pragma unroll
                for (int t = 0; t < 8; t++) ph *= (j0 + t < Mh) ? f[t] : 1.0;
              }
              return ph;
            };
            // Which haplotypes does the lane's request change?  Mutation proposals change one, and then the sub-groups
            // -- each with a different changed haplotype -- form their product in ONE pass of the whole wavefront instead
            // of one divergent pass per haplotype.  Same factors, same order of the sum over haplotypes: same values.
            int hc = -1, nchg = 0;
            uint64_t wac = 0;
            if (sub < nq) {
#pragma unroll
              for (int h = 0; h < KT; h++) {
                const uint64_t wa = pwbuf[(size_t)h * WAVE + my_src];
                if (wa != bw_tab[(size_t)sg * KT + h]) {
                  if (nchg == 0) {
                    hc = h;
                    wac = wa;
                  }
                  nchg++;
                }
              }
            }
            double sv = 0.0;
            if (!wave_any(nchg > 1)) {
              const double pc = nchg == 1 ? hap_product(wac) : 1.0;
              if (sub < nq) {
                double acc = 0.0;
#pragma unroll
                for (int h = 0; h < KT; h++) {
                  const double ph = (h == hc) ? pc : bpc[(h * 4) * WAVE + r];
                  acc += ph * invK;
                }
                sv = read_log(acc) * cwr;
              }
            } else if (sub < nq) {
              double acc = 0.0;
#pragma unroll
              for (int h = 0; h < KT; h++) {
                const uint64_t wa = pwbuf[(size_t)h * WAVE + my_src], wb = bw_tab[(size_t)sg * KT + h];
                const double ph = (wa == wb) ? bpc[(h * 4) * WAVE + r] : hap_product(wa);
                acc += ph * invK;
              }
              sv = read_log(acc) * cwr;
            }
            for (int o = RS / 2; o >= 1; o >>= 1) sv += __shfl_xor(sv, o, WAVE);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const double vk = __shfl(sv, (k * RS) & (WAVE - 1), WAVE);
              if (k < nq && ((dupm[k] >> lane) & 1ull)) val = vk;
            }
          }
          continue;  // next chain
        }
        // 33 to 64 reads, several requests (a side-by-side pass costs more than one plain evaluation -- three or four reads
        // per lane --, so a lone request takes the plain path below)
        if (use_base && nch == 1 && (reqs & (reqs - 1ull)) != 0ull) {
          // Units of one chunk (at most 64 reads): FOUR requests are evaluated side by side, a sub-group of 16 lanes each,
          // lane r of a sub-group taking the reads r, r + 16, r + 32, r + 48 of the unit (RK = 1..4 of them exist).  A
          // lane forms the same factors in the same order for its (request, read) as the plain path; the sum over the
          // reads is the wavefront butterfly (wave_sum: partners at distance 32, 16, 8, ... 1): the lane adds its reads
          // r and r + 32, and r + 16 and r + 48 (the step at distance 32), then those two sums (distance 16), and the
          // distances 8 .. 1 stay inside the sub-group.  Reads beyond the unit's are padding -- weight 0, a term of
          // +-0.0, which every sum absorbs exactly -- and are left out.  Bit for bit the values of the one-request path.
          const int RK = (nrd + 15) >> 4;
          const int sub = lane >> 4, r = lane & 15;
          LDSP(double) dict = S.dict + (size_t)sg * DICT_MAX;
          LDSP(const uint8_t) shift = shift_tab + (size_t)sg * mmax;
          LDSP(const uint16_t) cols = cols_tab + (size_t)sg * mmax;
          GLBP(const uint8_t) ctb = (GLBP(const uint8_t))(uintptr_t)gp[GP_CT];
          GLBP(const double) cwg = (GLBP(const double))(uintptr_t)gp[GP_CW];
          double cwr[4];
#pragma unroll
          for (int k = 0; k < 4; k++) cwr[k] = k < RK ? cwg[r + 16 * k] : 0.0;
          const double invK = 1.0 / (double)KT;
          // the products over the positions of one haplotype word for the lane's reads, four positions at a time: the code
          // loads of all reads go out together, then the dictionary gathers, then the products in position order (a
          // position beyond Mh multiplies by 1.0: exact) -- one memory round trip per four positions, not one per factor
          auto hap_products = [&](uint64_t wa, double (&ph)[4]) {
#pragma unroll
            for (int k = 0; k < 4; k++) ph[k] = 1.0;
            for (int j0 = 0; j0 < Mh; j0 += 4) {
              uint32_t row[4];
#pragma unroll
              for (int t = 0; t < 4; t++) {
                const int j = min(j0 + t, Mh - 1);
                row[t] = (uint32_t)cols[j] + ((uint32_t)(wa >> shift[j]) & amask);
              }
              uint8_t cd[4][4];
#pragma unroll
              for (int k = 0; k < 4; k++) {
                if (k < RK) {  // (wave-uniform)
                  if (sct != nullptr) {  // the unit's codes are in LDS: no memory round trip at all
#pragma unroll
                    for (int t = 0; t < 4; t++) cd[k][t] = sct[sct_off(row[t]) + r + 16 * k];
                  } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) cd[k][t] = ctb[(size_t)(row[t] * WAVE + r + 16 * k) * cstride];
                  }
                }
              }
#pragma unroll
              for (int k = 0; k < 4; k++) {
                if (k < RK) {
                  double f[4];
#pragma unroll
                  for (int t = 0; t < 4; t++) f[t] = dict[cd[k][t]];
#pragma unroll
                  for (int t = 0; t < 4; t++) ph[k] *= (j0 + t < Mh) ? f[t] : 1.0;
                }
              }
            }
          };
          while (reqs) {
            int srcs[4] = {0, 0, 0, 0};
            unsigned long long dupm[4] = {0ull, 0ull, 0ull, 0ull};
            int nq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (reqs) {
                const int src = __ffsll((long long)reqs) - 1;
                const unsigned long long dups = (DEDUP ? spec_same_request<KT>(pwbuf, reqs, src, lane) : (1ull << src));
                reqs &= ~dups;
                srcs[k] = src;
                dupm[k] = dups;
                nq = k + 1;
              }
            }
            const int my_src = sub == 0 ? srcs[0] : (sub == 1 ? srcs[1] : (sub == 2 ? srcs[2] : srcs[3]));
            // Which haplotypes does the lane's request change?  Mutation proposals change one, and then the sub-groups
            // -- each with a different changed haplotype -- form their products in ONE pass of the whole wavefront instead
            // of one divergent pass per haplotype.  Same factors, same order of the sum over haplotypes: same values.
            int hc = -1, nchg = 0;
            uint64_t wac = 0;
            if (sub < nq) {
#pragma unroll
              for (int h = 0; h < KT; h++) {
                const uint64_t wa = pwbuf[(size_t)h * WAVE + my_src];
                if (wa != bw_tab[(size_t)sg * KT + h]) {
                  if (nchg == 0) {
                    hc = h;
                    wac = wa;
                  }
                  nchg++;
                }
              }
            }
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            if (!wave_any(nchg > 1)) {
              double pc[4] = {1.0, 1.0, 1.0, 1.0};
              if (nchg == 1) hap_products(wac, pc);
              if (sub < nq) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  if (k < RK) {
                    double acc = 0.0;
#pragma unroll
                    for (int h = 0; h < KT; h++) {
                      const double ph = (h == hc) ? pc[k] : bpc[(h * 4) * WAVE + r + 16 * k];
                      acc += ph * invK;
                    }
                    v[k] = read_log(acc) * cwr[k];
                  }
                }
              }
            } else if (sub < nq) {
              double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int h = 0; h < KT; h++) {
                const uint64_t wa = pwbuf[(size_t)h * WAVE + my_src], wb = bw_tab[(size_t)sg * KT + h];
                double ph[4];
                if (wa == wb) {
#pragma unroll
                  for (int k = 0; k < 4; k++) ph[k] = k < RK ? bpc[(h * 4) * WAVE + r + 16 * k] : 0.0;
                } else {
                  hap_products(wa, ph);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] += ph[k] * invK;
              }
#pragma unroll
              for (int k = 0; k < 4; k++)
                if (k < RK) v[k] = read_log(acc[k]) * cwr[k];
            }
            // distance 32: reads r | r + 32 and r + 16 | r + 48; distance 16: the two sums; then inside the sub-group
            const double a0 = RK > 2 ? v[0] + v[2] : v[0];
            const double a1 = RK > 3 ? v[1] + v[3] : v[1];
            double sv = RK > 1 ? a0 + a1 : a0;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) sv += __shfl_xor(sv, o, WAVE);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const double vk = __shfl(sv, 16 * k, WAVE);
              if (k < nq && ((dupm[k] >> lane) & 1ull)) val = vk;
            }
          }
          continue;  // next chain
        }
      }
#endif
      while (reqs) {
        const int src = __ffsll((long long)reqs) - 1;
        // requests for the same genotype (options of different intervals often coincide) are evaluated once
        const unsigned long long dups = (DEDUP ? spec_same_request<KT>(pwbuf, reqs, src, lane) : (1ull << src));
        reqs &= ~dups;
        double s = 0.0;
        if (nb0 == 1) s += spec_coop_reuse<KT, 1, uint8_t, LT>(S, src, sg, mmax, Mh, amask, ct, cw, crow, lane, bp, use_base, grouped);
        else if (nb0 == 2) s += spec_coop_reuse<KT, 2, uint16_t, LT>(S, src, sg, mmax, Mh, amask, ct, cw, crow, lane, bp, use_base, grouped);
        else if (nb0 == 3) s += spec_coop_reuse<KT, 3, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct, cw, crow, lane, bp, use_base, grouped);
        else s += spec_coop_reuse<KT, 4, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct, cw, crow, lane, bp, use_base, grouped);
        for (int cb = 4; cb < nch; cb += 4) {
          const int rem = nch - cb;
          typename TabPtr<LT>::f64 cwb = cw + cb * WAVE;
          if (deep) {  // (wave-uniform) unchanged haplotypes: their products from the chain's rows
            BaseProductsG<KT> bg;
            bg.p = gbp;
            bg.rpad = rpad;
            bg.cb = cb;
            if (rem >= 4) s += spec_coop_reuse_g<KT, 4, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, bg, grouped);
            else if (rem == 3) s += spec_coop_reuse_g<KT, 3, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, bg, grouped);
            else if (rem == 2) s += spec_coop_reuse_g<KT, 2, uint16_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, bg, grouped);
            else s += spec_coop_reuse_g<KT, 1, uint8_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, bg, grouped);
          } else {
            if (rem >= 4) s += spec_coop_coded<KT, 4, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else if (rem == 3) s += spec_coop_coded<KT, 3, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else if (rem == 2) s += spec_coop_coded<KT, 2, uint16_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else s += spec_coop_coded<KT, 1, uint8_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
          }
        }
        s = wave_sum(s);
        if ((dups >> lane) & 1ull) val = s;
      }
    } else {
      while (reqs) {
        const int src = __ffsll((long long)reqs) - 1;
        const unsigned long long dups = (DEDUP ? spec_same_request<KT>(pwbuf, reqs, src, lane) : (1ull << src));
        reqs &= ~dups;
        // the lane's reads in blocks of at most 4 chunks of 64 (keeps the loads in flight, and the registers,
        // bounded whatever the read depth); the last block may hold 1-3 chunks
        double s = 0.0;
        const bool coded = nd_count(ndict_tab[sg]) != 0;
        typename TabPtr<LT>::u8 ct;
        if constexpr (LT) ct = lds_ct + (size_t)lane * cstride;
        else ct = (GLBP(const uint8_t))(uintptr_t)gp[GP_CT] + (size_t)lane * cstride;
        GLBP(const double) cwg = (GLBP(const double))(uintptr_t)gp[GP_CW] + lane;  // the float64 rows go with the global weights
        for (int cb = 0; cb < nch; cb += 4) {
          const int rem = nch - cb;
          typename TabPtr<LT>::f64 cwb = cw + cb * WAVE;
          if (coded) {
            if (rem >= 4) s += spec_coop_coded<KT, 4, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else if (rem == 3) s += spec_coop_coded<KT, 3, uint32_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else if (rem == 2) s += spec_coop_coded<KT, 2, uint16_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
            else s += spec_coop_coded<KT, 1, uint8_t, LT>(S, src, sg, mmax, Mh, amask, ct + cb, cwb, crow, lane, grouped);
          } else {
            GLBP(const double) rtb = rt + cb * WAVE;
            GLBP(const double) cwgb = cwg + cb * WAVE;
            if (rem >= 4) s += spec_coop_body<KT, 4>(S, src, sg, mmax, Mh, amask, rtb, cwgb, rpad, lane, nrd - cb * WAVE, grouped);
            else if (rem == 3) s += spec_coop_body<KT, 3>(S, src, sg, mmax, Mh, amask, rtb, cwgb, rpad, lane, nrd - cb * WAVE, grouped);
            else if (rem == 2) s += spec_coop_body<KT, 2>(S, src, sg, mmax, Mh, amask, rtb, cwgb, rpad, lane, nrd - cb * WAVE, grouped);
            else s += spec_coop_body<KT, 1>(S, src, sg, mmax, Mh, amask, rtb, cwgb, rpad, lane, nrd - cb * WAVE, grouped);
          }
        }
        s = wave_sum(s);
        if ((dups >> lane) & 1ull) val = s;
      }
    }
  }
  return val;
}

// Likelihood of the lane's proposal `pw` (where need): 4-way cache probe, then co-operative evaluation of the
// misses by the whole wavefront (request words staged through LDS).  Every lane of the wave must call.
// `filt(miss, val)` is asked once, by every lane, between the look-up and the evaluations: it may withdraw a lane's miss (the
// caller then gets no value for that lane and must not use one) -- spec_mutation drops the proposals behind a sub-step that is
// already known to move.
struct SpecKeepAll {
  __device__ __forceinline__ bool operator()(bool miss, double) const { return miss; }
};
template <int KT, int G, bool LT = false, class Filter = SpecKeepAll, bool DEDUP = true>
__device__ __forceinline__ double spec_eval(bool need, const GWords<KT> pw, const Grp<KT> &c, const SpecLds &S, int mmax,
                                            int rpad, int lane, Filter filt = Filter()) {
  double val = 0.0;
  // A unit without information (a sample with no reads at the locus: one all-gap row, every factor 1.0) gives every
  // genotype the same likelihood -- the same arithmetic on the same values --, which the chain already holds: no
  // probe, no evaluation.  (Such chains move at every other sub-step; they used to be the slowest of a launch.)
  if (c.flat) {
    val = c.llk;
    need = false;
  }
  bool miss = need;
  uint64_t tag = 0;
  ulonglong2 *slot = nullptr;
  GSUB_T0();
  // Genotypes wider than 63 bits: the tag is a hash, so a tag match is verified against the genotype's words kept
  // beside the entry (SimtParams::d.cache_keys) -- a hit is always the genotype itself, never a colliding one.
  const bool wide = C_KEYBITS(c) * KT > 63;
  uint64_t *kslot = nullptr;
  LDSP(uint64_t) lslot = S.lc;  // (only used where lc_mask != 0)
  bool promote = false;
  if (need && S.cache_on) {
    // 8-way sets (one 128-byte line).  The lanes of a group probe and fill the chain's table concurrently, so the
    // policy must not depend on read-modify-write sequences: hits never move entries, a miss goes to the first
    // empty way it saw (else to a way picked by the key); a lost race only costs one more evaluation later.
    tag = tag_of<KT>(pw, C_KEYBITS(c), S.epoch);
    const uint64_t key = tag >> 1;
    // full-avalanche 32-bit mix: the keys probed together are single-field neighbours of one genotype, so a
    // plain multiplicative hash would send all neighbours that differ in a high field to the same set
    uint32_t hsh = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    hsh ^= hsh >> 16;
    hsh *= 0x7FEB352Du;
    hsh ^= hsh >> 15;
    hsh *= 0x846CA68Bu;
    hsh ^= hsh >> 16;
    const size_t set_i = (size_t)((hsh >> 12) & S.cache_mask);
    ulonglong2 *set = reinterpret_cast<ulonglong2 *>((uintptr_t)S.gptr[(lane / G) * GP_N + GP_CACHE]) + 8 * set_i;
    int way = (int)((hsh >> 24) & 7u);
    if (S.lc_mask != 0u && !wide) {
      // Front cache in LDS (round 4; one chain per wave, exact tags): a probe of the table in the workspace is a memory round
      // trip of ~1.2 us in EVERY speculation round, about as long as the round's evaluations.  The LDS table is authoritative
      // for the chain's own steps -- a miss is evaluated, which is results-neutral like every miss --; what is evaluated still
      // goes to the table in the workspace (a blind store to the way the key picks: nothing waits for it), where the table
      // completion's launch (denovo_fillw_kernel.hpp) looks it up.
      LDSP(uint64_t) le = S.lc + 2 * (size_t)((hsh >> 3) & S.lc_mask);
      const uint64_t t = le[0], v = le[1];
      if (t == tag) {
        val = __longlong_as_double((long long)v);
        miss = false;
      }
      lslot = le;
      // A chain that keeps moving (dozens of genotypes behind it: samples with few reads) revisits more genotypes than 256
      // entries hold; its front-cache misses go on to the 1024-entry table in the workspace (whose latency such a chain pays
      // anyway: a round of it is mostly evaluations).  A settling chain -- BASELINE configs[1]: ~15 moves -- never gets here.
      if (miss && c.gen > SPEC_LC_SECOND_LEVEL_GEN) {
#pragma unroll
        for (int w = 7; w >= 0; w--) {
          const ulonglong2 e = set[w];
          if (e.x == tag) {
            val = __longlong_as_double((long long)e.y);
            miss = false;
          }
        }
        promote = !miss;  // (the next visit hits in LDS: written below, tag first, as after an evaluation)
      }
    } else {
    int hit_way = -1;
#pragma unroll
    for (int w = 7; w >= 0; w--) {
      const ulonglong2 e = set[w];
      if (e.x == tag) {
        val = __longlong_as_double((long long)e.y);
        miss = false;
        hit_way = w;
      }
      if (e.x == 0ull || (S.epoch != 0ull && (e.x ^ S.epoch) >> 33 != 0ull)) way = w;  // lowest empty way wins (an earlier call's entry is empty too)
    }
    if (wide) {
      uint64_t *kset = reinterpret_cast<uint64_t *>((uintptr_t)S.gptr[(lane / G) * GP_N + GP_CKEYS]) + 8 * set_i * S.key_words;
      if (hit_way >= 0) {
        const uint64_t *kw = kset + (size_t)hit_way * S.key_words;
        bool same = true;
#pragma unroll
        for (int h = 0; h < KT; h++) same = same & (kw[h] == pw.w[h]);
        if (!same) {  // another genotype with this tag: evaluate, and take over its entry (one entry per tag and set)
          miss = true;
          way = hit_way;
        }
      }
      kslot = kset + (size_t)way * S.key_words;
    }
    }
    slot = set + way;
  }
  if (wave_any(promote)) {
    if (promote) lslot[0] = tag;
    lds_sync();
    if (promote && lslot[0] == tag) lslot[1] = (uint64_t)__double_as_longlong(val);
    lds_sync();
  }
  miss = filt(miss, val);
  STAT_ADD(0, need);
  STAT_ADD(1, miss);
  STAT_ADD(2, true);
  unsigned long long todo = __ballot(miss);
  GSUB(c, 9);
  if (todo) {
    GCOUNT(c, 11, __popcll(todo));
    if (miss) {
#pragma unroll
      for (int h = 0; h < KT; h++) S.pw[(size_t)h * WAVE + lane] = pw.w[h];
    }
    if (S.reuse_on && lane % G == 0) {  // the chain's current genotype: what its proposals are variations of
      const GWords<KT> cg = c.g;
#pragma unroll
      for (int h = 0; h < KT; h++) S.bw[(size_t)(lane / G) * KT + h] = cg.w[h];
    }
    lds_sync();
    const double v = spec_coop_all<KT, G, LT, DEDUP>(todo, S.pw, S.shift, S.cols, S.nreads, S.ndict, S.dict, S.gptr, S.bw, S.reuse_on, S.crow, mmax, c.Mh, C_AMASK(c), rpad, lane, S.lds_ct, S.lds_cw, S.bpc, S.bpt, S.sct, S.gbt);
    if (miss) val = v;
    bool writer = miss && slot != nullptr;
    if (wave_any(writer && wide)) {
      // wide entries are two stores (words, then tag + value): of the lanes that picked the same way only the first
      // writes, so that an entry's words always belong to its tag
      unsigned long long pend = __ballot(writer);
      bool mine = false;
      while (pend) {
        const int l = __ffsll((long long)pend) - 1;
        const unsigned long long sl = __shfl((unsigned long long)(uintptr_t)slot, l, WAVE);
        const unsigned long long same = __ballot(writer && (unsigned long long)(uintptr_t)slot == sl);
        if (lane == l) mine = true;
        pend &= ~same;
      }
      writer = mine;
    }
    if (writer) {
      if (wide) {
#pragma unroll
        for (int h = 0; h < KT; h++) kslot[h] = pw.w[h];
      }
      *slot = make_ulonglong2(tag, (unsigned long long)__double_as_longlong(val));
    }
    if (S.lc_mask != 0u && !wide) {
      // the front cache: lanes with different keys may pick the same entry in one round -- every lane writes its tag, the one
      // whose tag stayed writes the value, so an entry's value always belongs to its tag
      if (writer) lslot[0] = tag;
      lds_sync();
      if (writer && lslot[0] == tag) lslot[1] = (uint64_t)__double_as_longlong(val);
    }
    lds_sync();
    GSUB(c, 10);
  }
  return val;
}

// mutation.compound_step (mutation.py:164-246) for the group's chain: shuffle, then speculative sub-steps.
#ifndef MCHAP_SPEC_WPE
#define MCHAP_SPEC_WPE 2
#endif

template <int KT, int G, bool LT = false, bool CTX = false>
__device__ __forceinline__ void spec_mutation(Grp<KT> &c, const SpecLds &S, double temp, int amax, int mmax, int nmax, int rpad,
                                              int lane, int gi, int gl) {
  const int Mh = c.Mh;
  const int n = KT * Mh;  // sub-steps; position p of the sequence is held by lane p % G, slot p / G (n <= NS G)
  const uint64_t ctr0 = c.ctr;
  // Fast path.  For an unchanged genotype the outcome of sub-step e depends only on its uniform: it stays iff
  // lo_e <= u < hi_e (the cumulative probabilities around the current allele).  With mlo = max lo_e and
  // mhi = min hi_e remembered from the last full evaluation, a compound step whose n uniforms all fall in
  // [mlo, mhi) moves nothing whatever the shuffle pairs them with: its 2n-1 draws are skipped over.
  bool run = c.alive;
  if (wave_any(c.alive && c.mvalid)) {
    LDSP(uint64_t) utab = S.draws + gi * S.ndraws;
    const bool fast = c.alive && c.mvalid && n <= S.ndraws;
    // the n uniforms, plus as many of the following draws as the same number of Philox rounds per lane yields:
    // they are the structural steps' draws if this step moves nothing
    const int cnt = max(n, min(S.ndraws, 2 * G * ((n / 2 + G) / G) - 1));
    stage_draws<G>(ld_stream(S, gi), ctr0 + (uint64_t)(n - 1), cnt, utab, gl, fast);
    lds_sync();
    bool ok = true;
    if (fast) {
      const double mlo = S.gval[gi * GV_N + GV_MLO], mhi = S.gval[gi * GV_N + GV_MHI];
      for (int p = gl; p < n; p += G) {
        const double u = draw_double(utab[p]);
        ok = ok && (mlo <= u) && (u < mhi);
      }
    }
    const bool bad = grp_ballot<G>(!ok, gi) != 0ull;
    lds_sync();
    if (fast && !bad) {
      c.ctr = ctr0 + (uint64_t)(n - 1) + (uint64_t)n;
      c.doff = n;
      c.dcount = cnt;
      run = false;
    }
  }
  if (run) c.dcount = 0;  // the slow path restages the table from ctr0 and uses all of it
  STAT_WAVE(8, 1);
  GPHASE(c, 0);
  if (!wave_any(run)) return;
  STAT_WAVE(9, 1);
  STAT_ADD(10, run && gl == 0);
  GT0(t_mut0);
  LDSP(uint8_t) ktab = S.ktab + gi * nmax;
  LDSP(uint16_t) permtab = S.permtab + gi * nmax;
  LDSP(uint8_t) shift = S.shift + gi * mmax;
  LDSP(uint8_t) nal = S.nal + gi * mmax;
  LDSP(double) pt = S.prior + gi * (2 * KT + 5);
  // sub-steps per lane ("slots"): two; three with one chain per wavefront, i.e. up to 3 G = 192 sub-steps (octoploids with
  // 20 SNVs and more, tetraploids with 33-48).  Only the slots in use are visited (nslots below).
  constexpr int NS = (G == 64) ? 3 : 2;
  const bool two = wave_any(c.alive && n > G);  // second slot in use anywhere in the wave
  // (1) the 2n-1 draws of this compound step, staged through LDS; the Fisher-Yates shuffle is swap(i, k_i) for
  //     i = n-1 .. 1 with k_i = interval(i) from draw ctr0 + (n-1-i)
  LDSP(uint64_t) dtab = S.draws + gi * S.ndraws;
  const bool staged = 2 * n - 1 <= S.ndraws;
  stage_draws<G>(ld_stream(S, gi), ctr0, 2 * n - 1, dtab, gl, run && staged);
  lds_sync();
  int kv[NS];  // k_p of the lane's positions p = gl + s G
#pragma unroll
  for (int s = 0; s < NS; s++) kv[s] = 0;
  if (run) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int p = gl + s * G;
      if (p >= 1 && p < n) {
        kv[s] = (int)(staged ? draw_interval(dtab[n - 1 - p], (uint32_t)p)
                             : stream_interval(ld_stream(S, gi), ctr0 + (uint64_t)(n - 1 - p), (uint32_t)p));
        if constexpr (G != 64) ktab[p] = (uint8_t)kv[s];
      }
    }
  }
  if constexpr (G != 64) lds_sync();
  // (2) every lane traces the element that starts at its position through the transpositions
  int xs[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) xs[s] = gl + s * G;
  if constexpr (G == 64) {
    // one chain per wave: k_i sits in lane i % 64 of kv[i / 64] and i is wave-uniform -- a v_readlane per transposition
    // instead of a dependent LDS read (the loop was 9 000 of the 11 000 cycles a slow-path step spends before its first
    // round at 32 sub-steps: profiles/r05d_phases_moving.txt)
    const int nl = __builtin_amdgcn_readfirstlane(run ? n : 0);
    if (nl <= 64) {  // one slot in use (configs[1]: 32 sub-steps): a third of the selects
      for (int i = nl - 1; i >= 1; i--) {
        const int ki = __builtin_amdgcn_readlane(kv[0], i);
        xs[0] = (xs[0] == i) ? ki : ((xs[0] == ki) ? i : xs[0]);
      }
    } else {
#pragma unroll
      for (int s2 = NS - 1; s2 >= 0; s2--) {
        const int hi = min(nl - 1, s2 * 64 + 63), lo = max(1, s2 * 64);
        for (int i = hi; i >= lo; i--) {
          const int ki = __builtin_amdgcn_readlane(kv[s2], i & 63);
#pragma unroll
          for (int s = 0; s < NS; s++) xs[s] = (xs[s] == i) ? ki : ((xs[s] == ki) ? i : xs[s]);
        }
      }
    }
  } else {
    const int nloop = run ? n : 0;
    int nl = nloop;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nl = max(nl, __shfl_xor(nl, o, WAVE));
    for (int i = nl - 1; i >= 1; i--) {
      if (i < nloop) {
        const int ki = ktab[i];
#pragma unroll
        for (int s = 0; s < NS; s++) xs[s] = (xs[s] == i) ? ki : ((xs[s] == ki) ? i : xs[s]);
      }
    }
  }
  if (run) {
    // element e = h * Mh + j starts at position e
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int e = gl + s * G;
      if (e < n) {
        int h = 0, j = e;
        while (j >= Mh) {
          j -= Mh;
          h++;
        }
        permtab[xs[s]] = (uint16_t)((h << 8) | j);
      }
    }
  }
  lds_sync();
  // (3) speculate / validate.  Position p = gl + s * G of the sequence is handled by slot s of lane gl; its
  //     sub-step is permtab[p] and its uniform is draw ctr0 + (n-1) + p.
  if (run) c.ctr = ctr0 + (uint64_t)(n - 1) + (uint64_t)n;
  const int nslots = (NS > 2 && wave_any(c.alive && n > 2 * G)) ? 3 : (two ? 2 : 1);
  // Window.  A round evaluates the sub-steps [start, start + win) only: while a chain is still converging nearly
  // every round ends at an accepted move and whatever was evaluated behind it is thrown away, so the window halves
  // after a round with a move and doubles after one without (Grp::mwin carries it to the next compound step; a
  // settled chain evaluates all n sub-steps in one round).  Results do not depend on it.
  // (a unit of at most 16 or of 33-64 reads gets four requests evaluated in one pass of the wavefront -- spec_coop_all --, so a
  // window of fewer than four sub-steps would leave that pass partly empty; 17-32 reads: two)
  const int nrd_ = (int)S.nreads[gi];
  const int wmin = (MCHAP_SPEC_SBS != 0 && G == 64 && S.bpc != nullptr && (nrd_ <= 16 || (nrd_ > 32 && nrd_ <= 64))) ? 4 : MCHAP_SPEC_WIN_MIN;
  // (wave-uniform where it matters: one chain per wave; groups of a shared wave all take the branch their wave takes)
  // decision contexts (CTX: one chain per wave, so everything here is wave-uniform): the sub-steps whose move probability the
  // current genotype's context holds decide themselves from it -- no proposal, no probe, no evaluation
  bool cx = false;
  if constexpr (CTX && G == 64) cx = S.cx_base != nullptr && amax == 2 && run && c.gen > (uint32_t)MCHAP_CTX_GEN && c.gen >= CXP_ST(S)[CX_PAUSE];
  const bool cut_on = amax == 2 && !S.cut_off && ((S.cache_on && wave_any(c.alive && c.gen > SPEC_LC_SECOND_LEVEL_GEN)) || cx);
  int start = 0;
  int win = max(c.mwin, wmin);
  bool done = !run;
  bool any_move = false;
  double my_lo = 0.0, my_hi = 2.0;  // this lane's sub-steps: max lo_e, min hi_e
  GT1(c, 12, t_mut0);
  while (wave_any(!done)) {
    STAT_WAVE(11, 1);
    if constexpr (CTX) STAT_WAVE(7, cx ? 1 : 0);
    bool found = false;
    int wstop = min(n, start + win);
    if constexpr (CTX && G == 64) {
      if (cx && !done) {
        // Known pass.  The sub-steps whose move probability the current genotype's context holds decide themselves: no proposal,
        // no option table, no exp().  The decisions are those of the general pass below for two alleles: with stay =
        // 1 - (0 + pr), a sub-step at allele 0 moves iff !(0 + stay > u), one at allele 1 iff 0 + pr > u.  fk: the first of
        // them that moves; fu: the first sub-step that is not in the context.
        ctx_acquire<KT>(c, S, n, mmax, lane);
        int fk = n, fu = n, m_h = 0;
        uint32_t m_w0 = 0, m_w1 = 0, m_l0 = 0, m_l1 = 0;
#pragma unroll 1
        for (int s = 0; s < nslots; s++) {
          const int p = gl + s * G;
          const bool in = p >= start && p < n;
          bool known = false, mv = false;
          int h = 0, sh = 0, current = 0;
          uint64_t wh = 0;
          double kllk = 0.0;
          if (in) {
            const int e = permtab[p];
            h = e >> 8;
            const int j = e & 255;
            const int ei = h * Mh + j;
            const double kpr = CXP_PR(S)[ei];
            known = kpr >= 0.0;
            if (known) {
              kllk = CXP_LLK(S, nmax)[ei];
              sh = shift[j];
              const double u = staged ? draw_double(dtab[n - 1 + p]) : stream_double(ld_stream(S, gi), ctr0 + (uint64_t)(n - 1) + (uint64_t)p);
              wh = sel_word<KT>(c.g, h);
              current = (int)((wh >> sh) & C_AMASK(c));
              const double stay = 1.0 - (0.0 + kpr);
              const double below = current == 0 ? 0.0 : 0.0 + kpr;  // cumulative probability below the current allele
              my_lo = fmax(my_lo, below);
              my_hi = fmin(my_hi, below + stay);
              mv = current == 0 ? !(0.0 + stay > u) : (0.0 + kpr > u);
            }
          }
          const unsigned long long km = __ballot(known), mm = __ballot(mv), um = __ballot(in && !known);
          STAT_WAVE(5, __popcll(km));
          STAT_WAVE(6, __popcll(um));
          if (mm && fk == n) {
            const int fl = __ffsll((long long)mm) - 1;
            fk = fl + s * G;
            const uint64_t nw_ = (wh & ~((uint64_t)C_AMASK(c) << sh)) | ((uint64_t)(1 - current) << sh);
            const long long lb = __double_as_longlong(kllk);
            m_h = __builtin_amdgcn_readlane(h, fl);
            m_w0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)nw_, fl);
            m_w1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(nw_ >> 32), fl);
            m_l0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lb, fl);
            m_l1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)lb >> 32), fl);
          }
          if (um && fu == n) fu = __ffsll((long long)um) - 1 + s * G;
        }
        if (fk < fu || fu >= n) {  // nothing unknown stands before the first known mover (or before the end of the step)
          if (fk < n) {
            set_word<KT>(c.g, m_h, (uint64_t)m_w0 | ((uint64_t)m_w1 << 32));
            c.llk = __longlong_as_double((long long)((uint64_t)m_l0 | ((uint64_t)m_l1 << 32)));
            genotype_changed<KT>(c);
            any_move = true;
            start = fk + 1;
          } else {
            start = n;
          }
          if (start >= n) done = true;
          continue;
        }
        // the general pass takes the unknown sub-steps of a window that counts from the first of them and ends at the first
        // known mover (the next round's known pass applies that one if nothing before it moves)
        wstop = min(fk, fu + win);
      }
    }
    const double lprior = (!done && !isnan(C_INB(S, gi))) ? prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(c.g)) : 0.0;
#pragma unroll 1
    for (int s = 0; s < nslots; s++) {
      const int p = gl + s * G;
      bool act = !done && !found && p >= start && p < wstop;
      int h = 0, sh = 0, n_alleles = 2;
      double u = 2.0;
      if (act) {
        const int e = permtab[p];
        h = e >> 8;
        const int j = e & 255;
        if constexpr (CTX && G == 64) {
          if (cx && CXP_PR(S)[h * Mh + j] >= 0.0) act = false;  // (the known pass took it)
        }
        sh = shift[j];
        n_alleles = nal[j];
        u = staged ? draw_double(dtab[n - 1 + p]) : stream_double(ld_stream(S, gi), ctr0 + (uint64_t)(n - 1) + (uint64_t)p);
      }
      const uint64_t wh = sel_word<KT>(c.g, h);
      const int current = (int)((wh >> sh) & C_AMASK(c));
      const double lhapcount = S.ln[copies_of<KT>(c.g, wh)];
      // options of the sub-step in allele order (current allele skipped); their move probabilities and likelihoods
      // are parked in the lane's LDS column so that the loop needs no unrolling
      const double ln_opt = S.ln[n_alleles - 1];
      double sum = 0.0;
      bool dropped = false;  // this lane's proposal was neither found nor evaluated: it lies behind a sub-step known to move
#pragma unroll 1
      for (int o = 0; o < amax - 1; o++) {  // wave-uniform trip count
        const bool prop = act && o < n_alleles - 1;
        const int i = o + (o >= current ? 1 : 0);
        GWords<KT> pw = c.g;
        const uint64_t nw = (wh & ~((uint64_t)C_AMASK(c) << sh)) | ((uint64_t)i << sh);
        set_word<KT>(pw, h, nw);
        // A chain with a history finds most of its proposals in its cache; the sub-steps whose proposal was found decide
        // themselves at once, and a miss BEHIND the first of them that moves cannot matter to this round -- it is not
        // evaluated (a chain that never settles evaluated four times what the sequential algorithm asks for: 18 per step at
        // docs/example's locus015 against the oracle's 4.4, profiles/r04m_*).  Biallelic sub-steps only (one proposal decides);
        // the look-ahead repeats the decision below operation for operation: stay = 1 - (0 + pr), the first cumulative value
        // above u chooses, beyond the last the reference clamps to the last allele.
        auto known_mover_cuts = [&](bool miss, double v) -> bool {
          if (!cut_on) return miss;  // (wave-uniform)
          bool mv = false;
          if (prop && !miss) {
            double lprior_ratio = 0.0;
            if (!isnan(C_INB(S, gi))) lprior_ratio = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(pw)) - lprior;
            const double lproposal_ratio = S.ln[copies_of<KT>(pw, nw)] - lhapcount;
            const double mh = ((v - c.llk) + lprior_ratio) * temp + lproposal_ratio;
            const double pr = exp(fmin(0.0, mh) - ln_opt);
            const double stay = 1.0 - (0.0 + pr);
            mv = current == 0 ? !(0.0 + stay > u) : (0.0 + pr > u);
          }
          const uint64_t mm = grp_ballot<G>(mv, gi);
          const bool keep = miss && (mm == 0ull || gl < __ffsll((long long)mm) - 1);
          dropped = dropped || (miss && !keep);
          return keep;
        };
        GT0(t_ev);
        const double llk_i = spec_eval<KT, G, LT, decltype(known_mover_cuts), G != 64>(prop, pw, c, S, mmax, rpad, lane, known_mover_cuts);
        GT1(c, 16, t_ev);
        if (prop) {
          double lprior_ratio = 0.0;
          if (!isnan(C_INB(S, gi))) lprior_ratio = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(pw)) - lprior;
          const double lproposal_ratio = S.ln[copies_of<KT>(pw, nw)] - lhapcount;
          const double mh = ((llk_i - c.llk) + lprior_ratio) * temp + lproposal_ratio;
          const double pr = exp(fmin(0.0, mh) - ln_opt);
          if constexpr (CTX && G == 64) {
            if (cx && !dropped) {  // into the current genotype's context: LDS, and written through to its slot
              const int e_ = permtab[p];
              const int ei = (e_ >> 8) * Mh + (e_ & 255);
              CXP_PR(S)[ei] = pr;
              CXP_LLK(S, nmax)[ei] = llk_i;
              uint64_t *sp = S.cx_base + (size_t)(CXP_ST(S)[CX_CUR] - 1u) * spec_ctx_words(KT, mmax) + SPEC_CTX_HDR;
              ctx_st(sp + ei, pr);
              ctx_st(sp + nmax + ei, llk_i);
            }
          }
          S.optp[o * WAVE + lane] = pr;
          S.optl[o * WAVE + lane] = llk_i;
          sum += pr;
        }
      }
      bool changed = false;
      uint64_t neww = 0;
      double newllk = 0.0;
      if (act && !dropped) {
        const double stay = 1.0 - sum;
        double cacc = 0.0;
        int choice = -1;
        double cl = c.llk;
        for (int i = 0; i < n_alleles && choice < 0; i++) {
          double pi = stay, li = c.llk;
          if (i != current) {
            const int o = i - (i > current ? 1 : 0);
            pi = S.optp[o * WAVE + lane];
            li = S.optl[o * WAVE + lane];
          } else {
            my_lo = fmax(my_lo, cacc);          // lo_e: cumulative probability below the current allele
            my_hi = fmin(my_hi, cacc + stay);   // hi_e: ... including it
          }
          cacc += pi;
          if (cacc > u) {
            choice = i;
            cl = li;
          }
        }
        if (choice < 0) {  // u beyond the last cumulative value: the reference's searchsorted returns n (clamped)
          choice = n_alleles - 1;
          if (choice != current) cl = S.optl[(choice - (choice > current ? 1 : 0)) * WAVE + lane];
        }
        if (choice != current) {
          changed = true;
          neww = (wh & ~((uint64_t)C_AMASK(c) << sh)) | ((uint64_t)choice << sh);
          newllk = cl;
        }
      }
      // first sub-step (in sequence order) that moves: positions of slot 0 precede those of slot 1
      const uint64_t m = grp_ballot<G>(changed, gi);
      const int fl = m ? __ffsll((long long)m) - 1 : 0;
      const int hsrc = __shfl(h, fl, G);
      const uint64_t wsrc = __shfl(neww, fl, G);
      const double lsrc = __shfl(newllk, fl, G);
      if (!done && !found && m) {
        set_word<KT>(c.g, hsrc, wsrc);
        c.llk = lsrc;
        genotype_changed<KT>(c);
        start = fl + s * G + 1;
        found = true;
      }
    }
    if (!done) {
      if (found) {
        any_move = true;
        win = max(wmin, win >> 1);
      } else {
        start = wstop;  // the window's sub-steps all stay
        win = min(2 * win, 4 * G);
      }
      if (start >= n) done = true;
    }
  }
  // a compound step without any move evaluated every sub-step against the current genotype: remember the bounds
  {
    double lo = my_lo, hi = my_hi;
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
      lo = fmax(lo, __shfl_xor(lo, o, G));
      hi = fmin(hi, __shfl_xor(hi, o, G));
    }
    if (run && !any_move && c.memo_on) {
      c.mvalid = true;
      if (gl == 0) {  // read again by the next mutation step, after several lds_sync()
        S.gval[gi * GV_N + GV_MLO] = lo;
        S.gval[gi * GV_N + GV_MHI] = hi;
      }
    }
  }
  if (run) c.mwin = win;
  GPHASE(c, 1);
}

// One structural compound step (structural.py:22-71, 433-673) of kind 0 recombination, 1 interval dosage,
// 2 whole-haplotype dosage.  Returns false if the group hit the reference's "breaks" ValueError.
// With PIPE, kind 3 / 4 is not a step: it completes the interval memo of step type 0 / 1 for the current genotype --
// every (start, stop) still unknown is evaluated by the same code (and so to the same totals) as a visit would, but
// without draws or decisions (phased sampler: the coasting kernel then decides every interval step from the table).
template <int KT, int G, bool PIPE = false, bool CTX = false>
__device__ __forceinline__ bool spec_structural(Grp<KT> &c, const SpecLds &S, const DenovoParams &D, int kind, double temp,
                                                const double *break_dist, int n_break_dist, int mmax, int rpad, int lane,
                                                int gi, int gl, int amax = 0) {
  const int Mh = c.Mh;
  const bool fill = PIPE && kind >= 3;  // wave-uniform
  const int step_type = (kind == 0 || kind == 3) ? 0 : 1;
  const int nivs = mmax + 1;
  LDSP(uint32_t) ivse = S.ivse + gi * nivs;
  LDSP(uint32_t) ivlin = S.ivlin + gi * nivs;
  LDSP(uint32_t) ivlout = S.ivlout + gi * nivs;
  LDSP(uint32_t) ivno = S.ivno + gi * nivs;
  LDSP(uint8_t) ord = S.ordtab + gi * nivs;
  LDSP(double) pt = S.prior + gi * (2 * KT + 5);
  bool ok = true;
  bool doit = false;
  int n_int = 0;
  // Draws come from the group's staged window (Grp::doff / dcount): table entry doff holds draw c.ctr.  The window
  // left behind by the mutation step usually covers all three structural steps; it is refilled (for every group
  // of the wave at once, one Philox block per lane) when a group runs low, and single draws beyond it are computed
  // on the spot.
  LDSP(uint64_t) dtab = S.draws + gi * S.ndraws;
  if (!fill) {
    const int low = min(3 * Mh + 2, MCHAP_SPEC_LOW);
    if (wave_any(c.alive && c.dcount - c.doff < low)) {
      const int W = min(S.ndraws, 2 * G - 1);
      lds_sync();
      stage_draws<G>(ld_stream(S, gi), c.ctr, W, dtab, gl, c.alive);
      lds_sync();
      c.doff = 0;
      c.dcount = c.alive ? W : 0;
    }
  }
  GPHASE(c, 2);
  // One chain per wave: the step's next 64 staged draws in a register, one per lane -- a sequential draw is then two v_readlane
  // instead of a dependent LDS read (a structural step makes 3 to 10 of them one after the other)
  const int dbase = G == 64 ? __builtin_amdgcn_readfirstlane(c.doff) : 0;
  uint64_t dreg = 0;
  if constexpr (G == 64) dreg = (!fill && dbase + lane < c.dcount) ? dtab[dbase + lane] : 0ull;
  auto next_words = [&]() -> uint64_t {
    const int i = c.doff;
    c.doff++;
    c.ctr++;
    if constexpr (G == 64) {
      const int k = __builtin_amdgcn_readfirstlane(i) - dbase;
      if (k >= 0 && k < WAVE && dbase + k < __builtin_amdgcn_readfirstlane(c.dcount)) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dreg, k);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dreg >> 32), k);
        return (uint64_t)lo | ((uint64_t)hi << 32);
      }
    }
    if (i < c.dcount) return dtab[i];
    uint32_t a, b;
    stream_words(ld_stream(S, gi), c.ctr - 1, a, b);
    return (uint64_t)a | ((uint64_t)b << 32);
  };
  uint64_t zeros = 0;
  if (c.alive && !fill) {
    const double pstep = kind == 0 ? D.p_recomb : (kind == 1 ? D.p_partial : D.p_dosage);
    doit = draw_double(next_words()) <= pstep;
    if (doit && kind < 2) {
      int nb;
      if (D.n_intervals > 0) {
        c.ctr++;  // break_dist = [0,...,0,1]: the draw is consumed (assemble/mcmc.py:214-217)
        c.doff++;
        nb = D.n_intervals - 1;
      } else {
        // first i with cumsum(break_dist)[i] > u (the cumulative sums were formed in the reference's order)
        const double u = draw_double(next_words());
        LDSP(double) bcum = S.bdist + gi * mmax;
        const uint64_t hit = grp_ballot<G>(gl < n_break_dist && bcum[gl < n_break_dist ? gl : 0] > u, gi);
        nb = hit ? __ffsll((long long)hit) - 1 : n_break_dist;
      }
      if (nb >= Mh) {
        ok = false;
        doit = false;
      } else {
        uint64_t ind = 0;
        for (int i = 1; i < Mh; i++) ind |= 1ull << i;
        for (int b = 0; b < nb; b++) {
          const int no = __popcll(ind);
          if (no == 0) break;
          int k = 0;
          if (no > 1) k = (int)draw_interval(next_words(), (uint32_t)(no - 1));
          uint64_t t = ind;
          while (k-- > 0) t &= t - 1;
          ind &= ~(t & (~t + 1));
        }
        zeros = ~ind & ((1ull << (Mh + 1)) - 1ull);
        n_int = nb + 1;
      }
    } else if (doit) {
      zeros = 1ull | (1ull << Mh);
      n_int = 1;
    }
  }
  GPHASE(c, 3);
  const uint64_t full = mask_of(c.bits, Mh, 0, Mh);
  // Memo.  For an unchanged genotype an interval step (type, start, stop) has a fixed option count and a fixed
  // total move probability (the last cumulative sum of its options): it moves nothing iff its uniform is >= that
  // total.  The table describes genotype generation memo_gen and is wiped when the genotype has changed.
  LDSP(double) mtot = S.memo_tot + gi * S.memo_stride + step_type * spec_memo_entries(mmax);
  const bool memo = PIPE || S.memo_stride != 0;  // (the phased form is only launched with the tables in place)
  // (CTX: one chain per wave) a chain with contexts takes the table of a genotype it has held before from that genotype's slot
  bool cx = false;
  if constexpr (CTX && G == 64) cx = S.cx_base != nullptr && amax == 2 && c.alive && c.gen > (uint32_t)MCHAP_CTX_GEN && c.gen >= CXP_ST(S)[CX_PAUSE];
  if (memo && wave_any(c.alive && c.gen != c.memo_gen)) {
    bool loaded = false;
    if constexpr (CTX && G == 64) {
      if (cx && !ctx_acquire<KT>(c, S, KT * Mh, mmax, lane)) {
        const uint64_t *tp = S.cx_base + (size_t)(CXP_ST(S)[CX_CUR] - 1u) * spec_ctx_words(KT, mmax) + SPEC_CTX_HDR + 2 * KT * mmax;
        LDSP(double) all = S.memo_tot + gi * S.memo_stride;
        for (int i0 = 0; i0 < S.memo_stride; i0 += 4 * WAVE) {
          uint64_t v[4];
#pragma unroll
          for (int t = 0; t < 4; t++) v[t] = i0 + t * WAVE + lane < S.memo_stride ? ctx_ld(tp + i0 + t * WAVE + lane) : 0ull;
#pragma unroll
          for (int t = 0; t < 4; t++)
            if (i0 + t * WAVE + lane < S.memo_stride) all[i0 + t * WAVE + lane] = __longlong_as_double((long long)v[t]);
        }
        c.memo_gen = c.gen;
        loaded = true;
      }
    }
    if (!loaded && c.alive && c.gen != c.memo_gen) {
      LDSP(double) all = S.memo_tot + gi * S.memo_stride;
      for (int i = gl; i < S.memo_stride; i += G) all[i] = NAN;
      c.memo_gen = c.gen;
    }
    lds_sync();
  }
  // (CTX) where a total of the current genotype goes besides the table in LDS
  auto ctx_tot = [&](int idx, double v) {
    if constexpr (CTX && G == 64) {
      if (cx && CXP_ST(S)[CX_GEN] == c.gen && CXP_ST(S)[CX_CUR] != 0u)
        ctx_st(S.cx_base + (size_t)(CXP_ST(S)[CX_CUR] - 1u) * spec_ctx_words(KT, mmax) + SPEC_CTX_HDR + 2 * KT * mmax + step_type * spec_memo_entries(mmax) + idx, v);
    }
  };
  GPHASE(c, 4);
  bool done = !doit;
  // Single-interval steps (the whole-haplotype dosage step, or no break drawn): one memo entry and at most one
  // uniform decide; no interval list, ballots or reduction needed.
  if (memo && doit && n_int == 1) {
    const double tot = mtot[spec_memo_index(0, Mh)];
    if (tot < 0.0) {
      done = true;  // no options: no draw
    } else if (!isnan(tot)) {
      const int i = c.doff;
      uint64_t w;
      if (i < c.dcount) {
        w = dtab[i];
      } else {
        uint32_t a, b;
        stream_words(ld_stream(S, gi), c.ctr, a, b);
        w = (uint64_t)a | ((uint64_t)b << 32);
      }
      if (draw_double(w) >= tot) {
        c.ctr++;
        c.doff++;
        done = true;
      }
    }
  }
  // Fast check, independent of the visiting order: if every interval of this compound step is in the memo, the
  // step consumes n_int - 1 shuffle draws and one uniform per interval that has options; it moves nothing if all
  // those uniforms are >= the largest total among its intervals, whichever interval each of them is paired with.
  if (memo && wave_any(doit && !done)) {
    const bool mine = doit && !done && gl < n_int;
    double tot = -1.0;
    if (mine) {
      uint64_t z = zeros;
      for (int q = 0; q < gl; q++) z &= z - 1;
      const int start = __ffsll((long long)z) - 1;
      z &= z - 1;
      const int stop = __ffsll((long long)z) - 1;
      tot = mtot[spec_memo_index(start, stop)];
    }
    const uint64_t unknown = grp_ballot<G>(mine && isnan(tot), gi);
    const int n_cons = __popcll(grp_ballot<G>(mine && tot >= 0.0, gi));
    const double mx = grp_max_f64<G>((mine && tot >= 0.0) ? tot : -1.0);
    bool low = false;
    if (doit && gl < n_cons) {
      const int i = c.doff + (n_int - 1) + gl;
      uint64_t w;
      if (i < c.dcount) {
        w = dtab[i];
      } else {
        uint32_t a, b;
        stream_words(ld_stream(S, gi), c.ctr + (uint64_t)(n_int - 1 + gl), a, b);
        w = (uint64_t)a | ((uint64_t)b << 32);
      }
      low = !(draw_double(w) >= mx);
    }
    const uint64_t anylow = grp_ballot<G>(low, gi);
    if (doit && !done && !unknown && !anylow) {
      c.ctr += (uint64_t)(n_int - 1 + n_cons);
      c.doff += n_int - 1 + n_cons;
      done = true;
    }
  }
  GPHASE(c, 5);
  // Exact path: visiting order (np.random.permutation(arange(n_int))), then the intervals one after the other
  int ii0 = 0;
  if (wave_any(!done)) {
    if (!done && gl == 0)
      for (int i = 0; i < n_int; i++) ord[i] = (uint8_t)i;
    lds_sync();
    if (!done) {
      for (int i = n_int - 1; i >= 1; i--) {
        const int k = (int)draw_interval(next_words(), (uint32_t)i);
        const uint8_t a = ord[i], b = ord[k];
        lds_sync();
        if (gl == 0) {
          ord[i] = b;
          ord[k] = a;
        }
        lds_sync();
      }
      // n_int <= Mh may exceed G for G = 16/32: strided
      for (int q = gl; q < n_int; q += G) {
        const int iv = ord[q];
        uint64_t z = zeros;
        for (int r = 0; r < iv; r++) z &= z - 1;
        const int start = __ffsll((long long)z) - 1;
        z &= z - 1;
        const int stop = __ffsll((long long)z) - 1;
        ivse[q] = (uint32_t)start | ((uint32_t)stop << 8);
      }
    }
    lds_sync();
    // intervals are skipped while the memo says "no move"
    if (!done && memo) {
      while (ii0 < n_int) {
        const uint32_t se = ivse[ii0];
        const int idx = spec_memo_index((int)(se & 255u), (int)(se >> 8));
        const double tot = mtot[idx];
        if (isnan(tot)) break;  // not evaluated for this genotype yet
        if (tot >= 0.0) {       // -1: the step has no options and consumes no draw
          const int i = c.doff;
          uint64_t w;
          if (i < c.dcount) {
            w = dtab[i];
          } else {
            uint32_t a, b;
            stream_words(ld_stream(S, gi), c.ctr, a, b);
            w = (uint64_t)a | ((uint64_t)b << 32);
          }
          if (!(draw_double(w) >= tot)) break;  // this interval moves: evaluate it for real
          c.ctr++;
          c.doff++;
        }
        ii0++;
      }
      if (ii0 >= n_int) done = true;
    }
  }
  GPHASE(c, 6);
  STAT_WAVE(12 + 4 * (kind > 0), 1);
  STAT_ADD(13 + 4 * (kind > 0), doit && gl == 0);
  STAT_ADD(14 + 4 * (kind > 0), !done && gl == 0);
  if (wave_any(!done)) STAT_WAVE(20 + (kind > 0), 1);
  int fill_next = 0;  // fill mode: next entry of the table to look at
  // (a chain that has moved since its last mutation step is not settled: the coasting kernel hands it straight back,
  // so its table is not worth completing)
  const bool filling = fill && memo && c.alive && c.mvalid;
  do {
  if (fill) {
    // the next (up to Mmax + 1) entries that are still unknown become this pass's interval list
    int cnt = 0;
    if (filling && gl == 0) {
      const int n_entries = spec_memo_entries(Mh);
      int p = fill_next, stop = 1;
      while (spec_memo_entries(stop) <= p) stop++;
      int start = p - spec_memo_index(0, stop);
      while (p < n_entries && cnt < nivs) {
        if (isnan(mtot[p]) && (step_type * spec_memo_entries(mmax) + p) % c.fill_parts == c.fill_part)
          ivse[cnt++] = (uint32_t)start | ((uint32_t)stop << 8);
        p++;
        if (++start == stop) {
          start = 0;
          stop++;
        }
      }
      fill_next = p;
    }
    n_int = __shfl(cnt, 0, G);
    fill_next = __shfl(fill_next, 0, G);
    ii0 = 0;
    done = !(filling && n_int > 0);
    lds_sync();
  }
  while (wave_any(!done)) {
    STAT_WAVE(15 + 4 * (kind > 0), 1);
    // (a) labels and option counts of the next (up to G) intervals, one interval per lane
    if (!done) {
      for (int ii = ii0 + gl; ii < n_int && ii < ii0 + G; ii += G) {
        const uint32_t se = ivse[ii];
        const uint64_t min_ = mask_of(c.bits, Mh, (int)(se & 255u), (int)(se >> 8));
        const uint32_t lin = seg_labels<KT>(c.g, min_);
        const uint32_t lout = seg_labels<KT>(c.g, full & ~min_);
        ivlin[ii] = lin;
        ivlout[ii] = lout;
        ivno[ii] = (uint32_t)(step_type == 0 ? recombination_n_options(lin, lout, KT) : dosage_n_options(lin, lout, KT));
      }
    }
    lds_sync();
    // (b) slots: lane gl serves option (my_ii, my_o); intervals [ii0, ii1) fit into G slots
    int ii1 = ii0, my_ii = -1, my_o = 0, my_no = 0;
    if (!done) {
      int off = 0;
      while (ii1 < n_int && ii1 < ii0 + G) {
        const int no = (int)ivno[ii1];
        if (off + no > G && ii1 > ii0) break;
        if (gl >= off && gl < off + no) {
          my_ii = ii1;
          my_o = gl - off;
          my_no = no;
        }
        off += no;
        ii1++;
      }
    }
    // (c) evaluate my option
    const bool prop = !done && my_ii >= 0;
    const GWords<KT> cg = c.g;
    GWords<KT> pw = cg;
    uint32_t oin = 0, lout = 0;
    if (prop) {
      const uint32_t se = ivse[my_ii];
      const uint64_t min_ = mask_of(c.bits, Mh, (int)(se & 255u), (int)(se >> 8));
      const uint32_t lin = ivlin[my_ii];
      lout = ivlout[my_ii];
      // the my_o-th option in the reference's enumeration order (structural.py:121-178 / 240-307)
      const uint32_t hd = dosage_of_labels(lin, lout, KT, true);
      int cnt = 0;
      if (step_type == 0) {
#pragma unroll
        for (int h0 = 0; h0 < KT; h0++) {
#pragma unroll
          for (int h1 = h0 + 1; h1 < KT; h1++) {
            const bool valid = nib(hd, h0) != 0 && nib(hd, h1) != 0 && nib(lin, h0) != nib(lin, h1) && nib(lout, h0) != nib(lout, h1);
            if (valid) {
              if (cnt == my_o) {
                uint32_t o = nib_set(lin, h0, nib(lin, h1));
                oin = nib_set(o, h1, nib(lin, h0));
              }
              cnt++;
            }
          }
        }
      } else {
        const uint32_t sd = dosage_of_labels(lin, lout, KT, false);
#pragma unroll
        for (int h0 = 0; h0 < KT; h0++) {
#pragma unroll
          for (int h1 = 0; h1 < KT; h1++) {
            const bool valid = nib(hd, h0) != 0 && nib(sd, h0) != 1 && nib(sd, h1) != 0 && nib(lin, h0) != nib(lin, h1);
            if (valid) {
              if (cnt == my_o) oin = nib_set(lin, h0, nib(lin, h1));
              cnt++;
            }
          }
        }
      }
#pragma unroll
      for (int h = 0; h < KT; h++) pw.w[h] = (cg.w[h] & ~min_) | (sel_word<KT>(cg, (int)nib(oin, h)) & min_);
    }
    GT0(t_ev);
    const double llk_i = spec_eval<KT, G>(prop, pw, c, S, mmax, rpad, lane);
    GT1(c, 17, t_ev);
    if (prop) {
      double lprior_ratio = 0.0;
      if (!isnan(C_INB(S, gi)))
        lprior_ratio = prior_of<KT>(pt, C_INB(S, gi), dosage_of_labels(oin, lout, KT, true)) -
                       prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(c.g));
      const int n_return = step_type == 0 ? recombination_n_options(oin, lout, KT) : dosage_n_options(oin, lout, KT);
      const double lproposal_ratio = S.lninv[n_return] - S.lninv[my_no];
      const double mh = ((llk_i - c.llk) + lprior_ratio) * temp + lproposal_ratio;
      S.ptab[lane] = exp(fmin(0.0, mh) - S.ln[my_no]);
    }
    lds_sync();
    // (d) sequential validation walk over the intervals of this round
    int acc_gl = -1;
    if (!done && fill) {
      // the totals of this round's intervals: the sum a visit without a move would have formed
      int off = 0;
      for (int ii = ii0; ii < ii1; ii++) {
        const int no = (int)ivno[ii];
        double cacc = 0.0;
        for (int o = 0; o < no; o++) cacc += S.ptab[gi * G + off + o];
        if (gl == 0) {
          const uint32_t se = ivse[ii];
          mtot[spec_memo_index((int)(se & 255u), (int)(se >> 8))] = no > 0 ? cacc : -1.0;
          ctx_tot(spec_memo_index((int)(se & 255u), (int)(se >> 8)), no > 0 ? cacc : -1.0);
        }
        off += no;
      }
      ii0 = ii1;
    } else if (!done) {
      int off = 0;
      int ii = ii0;
      for (; ii < ii1; ii++) {
        const int no = (int)ivno[ii];
        double cacc = 0.0;
        if (no > 0) {
          const double u = draw_double(next_words());
          int choice = -1;
          for (int o = 0; o < no; o++) {
            cacc += S.ptab[gi * G + off + o];
            if (cacc > u) {
              choice = o;
              break;
            }
          }
          if (choice >= 0) {
            acc_gl = off + choice;
            ii++;
            break;
          }
        }
        if (memo && gl == 0 && c.gen == c.memo_gen) {  // evaluated in full without a move: remember the total
          const uint32_t se = ivse[ii];
          const int idx = spec_memo_index((int)(se & 255u), (int)(se >> 8));
          mtot[idx] = no > 0 ? cacc : -1.0;
          ctx_tot(idx, no > 0 ? cacc : -1.0);
        }
        off += no;
      }
      ii0 = ii;
    }
    // (e) apply the accepted move (if any): its words and llk come from the lane that evaluated it
    {
      const int src = acc_gl >= 0 ? acc_gl : 0;
      GWords<KT> nw;
#pragma unroll
      for (int h = 0; h < KT; h++) nw.w[h] = __shfl(pw.w[h], src, G);
      const double nl = __shfl(llk_i, src, G);
      if (acc_gl >= 0) {
        c.g = nw;
        c.llk = nl;
        genotype_changed<KT>(c);
      }
    }
    if (!done && ii0 >= n_int) done = true;
    lds_sync();
  }
  } while (fill && wave_any(filling && n_int > 0));
  GPHASE(c, 7);
  return ok;
}



// PIPE: the phased form (kernel 5).  A launch runs the chains of P.pipe_list for P.pipe_iters compound steps each,
// starting from scratch or (PIPE_RESUME) from their PipeState records, and (PIPE_EXPORT) leaves records + complete
// interval memos behind for denovo_coast_kernel.  Single temperature only.
// TW (round 4, temperature ladders: G = 64, not PIPE): ONE WAVEFRONT PER REPLICA.  The T replicas of a chain are independent within
// an MCMC step (assemble/mcmc.py:323-395: every replica's moves start from its own state of the previous step); only the swap
// attempts couple neighbours, in the order t = 1 .. T-1, each with the state its lower neighbour holds THEN (tempering.py:61-151).
// A workgroup of T wavefronts therefore runs the replicas' moves side by side -- every wavefront with the sampler's LDS layout of its
// own behind a small exchange area --, publishes (genotype, llk), and the swap chain runs through that area behind workgroup
// barriers: wavefront t attempts its swap with t-1 in turn.  Same draws, same arithmetic, same order of the swaps: same traces as
// the one-wavefront form, which walks the replicas one after the other (tests/test_gpu_denovo.py::test_priors_and_tempering).
constexpr int SPEC_TW_MAX = 8;  // replicas a workgroup takes (longer ladders run on one wavefront)
__host__ __device__ inline size_t spec_tw_exchange_bytes(int K, int T) { return (((size_t)8 * T * (K + 1) + 16) + 63) & ~(size_t)63; }

// CTX (round 5, PIPE with G = 64): the chain keeps decision contexts per genotype in SimtParams::ctx (see "decision contexts" above);
// its own instantiation, so that the launches without contexts -- the first phase of every chain -- keep their code and registers.
template <int KT, int G, bool PIPE = false, int VAR = MCHAP_SPEC_VAR, bool TW = false, bool CTX = false>  // (VAR only names the object: the code is selected by the macro)
__global__ __launch_bounds__(TW ? 64 * SPEC_TW_MAX : 64, TW ? 2 : MCHAP_SPEC_WPE) void denovo_spec_kernel(const SimtParams P) {
  extern __shared__ __align__(16) unsigned char smem_all[];
  constexpr int NG = 64 / G;
  const DenovoParams &D = P.d;
  const int lane = threadIdx.x & (WAVE - 1);
  const int tw = TW ? (int)(threadIdx.x / WAVE) : 0;  // TW: the replica (temperature) this wavefront runs
  // TW: [exchange area][wavefront 0's layout][wavefront 1's] ..; else the workgroup is one wavefront and smem its layout
  unsigned char *smem = smem_all + (TW ? spec_tw_exchange_bytes(KT, D.n_temps) + (size_t)tw * (size_t)P.tw_lds : 0);
  LDSP(uint64_t) xw = lds_cast<uint64_t>(smem_all);                                  // TW: [T][K] the replicas' genotypes
  LDSP(double) xllk = lds_cast<double>(smem_all + (size_t)8 * D.n_temps * KT);       //     [T] their log likelihoods
  LDSP(int) xdead = lds_cast<int>(smem_all + (size_t)8 * D.n_temps * (KT + 1));      //     a replica stopped on an error
  if (TW && threadIdx.x == 0) *xdead = 0;
  const int gi = lane / G, gl = lane % G;
  int n_list = 0;
  long long my_slot = (long long)blockIdx.x * NG + gi;  // the group's position in the launch's list of chains
  int parts_eff = 1, my_part = 0;
  const bool fillonly = PIPE && (P.pipe_mode & PIPE_FILLONLY);
  if constexpr (PIPE) {
    n_list = P.pipe_count ? *P.pipe_count : (int)((long long)P.n_units * D.chains);
    parts_eff = pipe_parts_eff(n_list, P.pipe_parts);
    if (fillonly) {
      if (parts_eff <= 1) return;  // the exporting launch completed the tables itself
      const int n_slots = n_list * parts_eff;
      const int n_waves = min(n_slots, (int)gridDim.x);
      if ((int)blockIdx.x >= n_waves) return;
      const long long v = (long long)gi * n_waves + blockIdx.x;
      my_slot = v < n_slots ? v / parts_eff : (long long)n_list;
      my_part = (int)(v % parts_eff);
    } else if (P.pipe_list) {
      // a list of handed-back chains is usually short: its chains are dealt out over as many wavefronts as there
      // are (one chain per wave while they last), because a wave serves its chains' evaluations one after the other
      const int n_waves = min(n_list, (int)gridDim.x);
      if ((int)blockIdx.x >= n_waves) return;
      my_slot = (long long)gi * n_waves + blockIdx.x;
    } else if ((long long)blockIdx.x * NG >= n_list) {
      return;  // the grid is sized for every chain
    }
  }
  const int T = PIPE ? 1 : D.n_temps;  // the phased form is single-temperature: the ladder code drops out
  const int Cn = D.chains, Sn = D.steps;
  const int mmax = P.max_pos, nmax = KT * P.max_pos;
  const int rpad = D.rpad;
  SpecLds S;
  {
    const int nopt = P.max_allele > 1 ? P.max_allele - 1 : 1;
    const int niv = mmax + 1;
    unsigned char *p = smem;
    S.pw = lds_cast<uint64_t>(p); p += (size_t)8 * KT * 64;
    S.wst = lds_cast<uint64_t>(p); p += (size_t)8 * NG * T * KT;
    S.llk_t = lds_cast<double>(p); p += (size_t)8 * NG * T;
    S.rngn = lds_cast<uint64_t>(p); p += (size_t)8 * NG * T;
    S.prior = lds_cast<double>(p); p += (size_t)8 * NG * (2 * KT + 5);
    S.ptab = lds_cast<double>(p); p += (size_t)8 * 64;
    S.optp = lds_cast<double>(p); p += (size_t)8 * nopt * 64;
    S.optl = lds_cast<double>(p); p += (size_t)8 * nopt * 64;
    S.ln = lds_cast<double>(p); p += (size_t)8 * SPEC_LN;
    S.lninv = lds_cast<double>(p); p += (size_t)8 * SPEC_LN;
    S.bdist = lds_cast<double>(p); p += (size_t)8 * NG * mmax;
    S.ivse = lds_cast<uint32_t>(p); p += (size_t)4 * NG * niv;
    S.ivlin = lds_cast<uint32_t>(p); p += (size_t)4 * NG * niv;
    S.ivlout = lds_cast<uint32_t>(p); p += (size_t)4 * NG * niv;
    S.ivno = lds_cast<uint32_t>(p); p += (size_t)4 * NG * niv;
    S.cols = lds_cast<uint16_t>(p); p += (size_t)2 * NG * mmax;
    S.permtab = lds_cast<uint16_t>(p); p += (size_t)2 * NG * nmax;
    S.shift = lds_cast<uint8_t>(p); p += (size_t)NG * mmax;
    S.nal = lds_cast<uint8_t>(p); p += (size_t)NG * mmax;
    S.ktab = lds_cast<uint8_t>(p); p += (size_t)NG * nmax;
    S.ordtab = lds_cast<uint8_t>(p); p += (size_t)NG * niv;
    p = smem + (((size_t)(p - smem) + 1) & ~(size_t)1);
    S.nreads = lds_cast<uint16_t>(p); p += (size_t)2 * NG;
    S.ndict = lds_cast<uint16_t>(p); p += (size_t)2 * NG;
    p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
    S.dict = lds_cast<double>(p); p += (size_t)8 * NG * DICT_MAX;
    S.gptr = lds_cast<uint64_t>(p); p += (size_t)8 * NG * GP_N;
    S.gval = lds_cast<double>(p); p += (size_t)8 * NG * GV_N;
    p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
    S.gstream = lds_cast<uint32_t>(p); p += (size_t)16 * NG;
    S.bw = lds_cast<uint64_t>(p); p += (size_t)8 * NG * KT;
    S.gbt = lds_cast<uint64_t>(p); p += (size_t)8 * NG * (KT + 1);
    for (int i = lane; i < NG * (KT + 1); i += WAVE) S.gbt[i] = 0ull;  // nothing in the chain's product rows yet
    S.tbuf = lds_cast<uint64_t>(p); p += (size_t)8 * NG * SPEC_TB * (KT + 1);
    p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
    S.ndraws = spec_draws(KT, mmax);
    S.draws = lds_cast<uint64_t>(p); p += (size_t)8 * NG * S.ndraws;
    S.memo_stride = (spec_memo_bytes(mmax, T, G) && !(P.flags & 2)) ? 2 * spec_memo_entries(mmax) : 0;
    S.memo_tot = lds_cast<double>(p);
    for (int i = lane; i < NG * S.memo_stride; i += WAVE) S.memo_tot[i] = NAN;  // nothing evaluated yet
    S.bpc = nullptr;
    S.bpt = nullptr;
    S.lc = lds_cast<uint64_t>(p);
    S.lc_mask = 0u;
    if (G == 64 && P.bp_cache) {
      p += spec_memo_bytes(mmax, T, G);
      p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
      S.bpc = lds_cast<double>(p); p += (size_t)8 * KT * 4 * 64;
      S.bpt = lds_cast<uint64_t>(p); p += (size_t)8 * (KT + 1);
      if (lane == 0) S.bpt[KT] = 0ull;
      if (P.bp_cache & 2) {  // the chain's likelihood cache in LDS (spec_eval), carved behind the product cache
        p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
        S.lc = lds_cast<uint64_t>(p);
        const int lce = spec_lc_entries(P.bp_cache);
        S.lc_mask = (uint32_t)lce - 1u;
        for (int i = lane; i < 2 * lce; i += WAVE) S.lc[i] = 0ull;  // (a tag is never 0: tag_of)
        p += spec_lc_bytes(lce);
      }
    } else {
      p += spec_memo_bytes(mmax, T, G);
    }
    if constexpr (CTX && G == 64 && PIPE) {
      if (P.ctx != nullptr && P.ctx_n > 0) {  // directory and current context behind everything else (the host sized the LDS for it)
        if (P.bp_cache & 8) {  // inside the product cache (spec_ctx_in_bpc)
          S.cx_lds = (LDSP(unsigned char))(S.bpc + (size_t)((KT - 1) * 4 + 1) * WAVE);
          S.cx_val = S.bpc + (size_t)((KT - 2) * 4 + 1) * WAVE;
        } else {
          p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
          S.cx_lds = lds_cast<unsigned char>(p); p += (size_t)12 * SPEC_CTX_MAX + 4 * CX_N;
          S.cx_val = lds_cast<double>(p); p += (size_t)16 * nmax;
        }
        S.cx_n = P.ctx_n < SPEC_CTX_MAX ? P.ctx_n : SPEC_CTX_MAX;
        if (lane < CX_N) CXP_ST(S)[lane] = 0u;
        if (lane < SPEC_CTX_MAX) {
          CXP_TAG(S)[lane] = 0ull;
          CXP_STAMP(S)[lane] = 0u;
        }
      }
    }
  }
  for (int i = lane; i < SPEC_LN; i += WAVE) {
    S.ln[i] = c_ln[i];
    S.lninv[i] = c_ln_inv[i];
  }
  long long q = my_slot;  // chain index of the group
  const long long n_chains = (long long)P.n_units * Cn;
  Grp<KT> c;
  c.alive = q < n_chains;
  bool listed = false;  // PIPE: the group holds a chain of the list (its record is written at the end)
  if constexpr (PIPE) {
    listed = q < n_list;
    c.alive = listed;
    if (listed && P.pipe_list) q = P.pipe_list[q];
    if (!listed) q = 0;
  }
  const bool resume = PIPE && (P.pipe_mode & PIPE_RESUME);
  int base = 0;  // first MCMC step of this launch (PIPE)
  const int u = c.alive ? (int)(q / Cn) : 0;
  const int chain = c.alive ? (int)(q % Cn) : 0;
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);
  if (c.alive && mi[META_I_STATUS] != MCHAP_UNIT_OK) c.alive = false;
  const int A = U.max_allele;
  c.Mh = c.alive ? mi[META_I_MH] : 1;
  c.bits = allele_bits(A);
  c.flat = c.alive && mi[META_I_FLAT] != 0 && !(P.flags & 128);
  if (gl == 0) {
    S.gval[gi * GV_N + GV_INB] = U.inbreeding;
    S.gval[gi * GV_N + GV_MLO] = 0.0;
    S.gval[gi * GV_N + GV_MHI] = 0.0;
    S.gstream[gi * 4 + 0] = (uint32_t)D.seed;
    S.gstream[gi * 4 + 1] = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
    S.gstream[gi * 4 + 2] = ((uint32_t)chain << 16) | 0u;
    S.gstream[gi * 4 + 3] = (uint32_t)U.stream_id;
  }
  S.reuse_on = !(P.flags & 8);
  S.cut_off = (P.flags & 262144) != 0;
  S.lds_ct = nullptr;
  S.lds_cw = nullptr;
  S.crow = WAVE * P.cstride;
  S.cache_on = D.cache_slots > 0 && !fillonly;
  S.cache_mask = S.cache_on ? (uint32_t)(D.cache_slots / 8) - 1u : 0u;  // sets of 8 ways
  S.epoch = P.cache_epoch;
  S.key_words = D.cache_key_words;
  if (gl == 0) {
    LDSP(uint64_t) gp = S.gptr + gi * GP_N;
    gp[GP_RT] = (uint64_t)(uintptr_t)(P.rt + (size_t)u * P.max_ma * rpad);
    gp[GP_CW] = (uint64_t)(uintptr_t)(P.cntw + (size_t)u * rpad);
    gp[GP_CT] = (uint64_t)(uintptr_t)(P.codes + (size_t)u * P.max_ma * WAVE * P.cstride);
    // (TW: the replicas of a chain run on different wavefronts AT THE SAME TIME, and an entry is two 8-byte accesses -- a reader
    // could pair one wavefront's tag with another's value; every replica therefore has a table of its own: the workspace holds
    // n_temps tables per chain for a ladder on this kernel.  One wavefront walking the replicas shares the first, as before.)
    const size_t qc = (size_t)q * (size_t)(PIPE ? 1 : D.n_temps) + (size_t)tw;
    gp[GP_CACHE] = S.cache_on ? (uint64_t)(uintptr_t)(reinterpret_cast<ulonglong2 *>(D.cache) + qc * (size_t)D.cache_slots) : 0ull;
    gp[GP_CKEYS] = (S.cache_on && D.cache_keys) ? (uint64_t)(uintptr_t)(D.cache_keys + qc * (size_t)D.cache_slots * D.cache_key_words) : 0ull;
    gp[GP_TRACE] = (uint64_t)(uintptr_t)(D.trace + U.trace_off + (size_t)chain * D.steps * KT);
    gp[GP_LLK] = (uint64_t)(uintptr_t)(D.llks + U.llk_off + (size_t)chain * D.steps);
    // (a ladder's replicas run side by side, TW: each its own rows -- the workspace holds n_temps sets per chain then)
    gp[GP_GBP] = P.gbp ? (uint64_t)(uintptr_t)(P.gbp + ((size_t)q * (PIPE ? 1 : D.n_temps) + tw) * P.max_ploidy * rpad) : 0ull;
  }
  c.ctr = 0;
  c.doff = 0;
  c.dcount = 0;
  c.llk = 0.0;
  c.memo_on = (T == 1) && !(P.flags & 1);
  c.mvalid = false;
  c.mwin = MCHAP_SPEC_WIN0;
  c.fill_part = fillonly ? my_part : 0;
  c.fill_parts = fillonly ? parts_eff : 1;
  c.gen = 1;
  c.memo_gen = 1;
  if constexpr (CTX && G == 64 && PIPE) {  // the chain's region: [chain][ctx_n slots][spec_ctx_words]
    if (S.cx_n > 0 && c.alive) S.cx_base = P.ctx + (size_t)q * (size_t)P.ctx_n * (size_t)spec_ctx_words(KT, mmax);
  }
  const int Mh = c.Mh;
  if (gl == 0) S.nreads[gi] = (uint16_t)(c.alive ? U.n_reads : 0);
  {
    const int nd = (c.alive && !(P.flags & 4)) ? mi[META_I_NDICT] : 0;
    if (gl == 0) S.ndict[gi] = (uint16_t)(nd | ((c.alive && mi[META_I_W01] != 0) ? (int)ND_W01 : 0));
    const double *du = P.dict + (size_t)u * DICT_MAX;
    for (int i = gl; i < nd; i += G) S.dict[(size_t)gi * DICT_MAX + i] = du[i];
  }
  S.sct = nullptr;
  if constexpr (G == 64 && MCHAP_SPEC_SBS != 0) {
    // A unit of one chunk (at most 64 reads: spec_coop_all evaluates its requests side by side) with a small table:
    // every lane's code of every row goes into LDS once per launch -- into the chunk slots 1..3 of every haplotype of the
    // product cache (sct_off), which a one-chunk unit never uses -- and its evaluations then make no memory round trip.
    const int rows = U.n_pos * A;
    if (S.bpc != nullptr && c.alive && U.n_reads <= 64 && rows <= 24 * KT && mi[META_I_NDICT] != 0 && !(P.flags & 4)) {
      LDSP(uint8_t) t = (LDSP(uint8_t))(S.bpc + WAVE);
      GLBP(const uint8_t) ct = (GLBP(const uint8_t))(P.codes + (size_t)u * P.max_ma * WAVE * P.cstride);
      for (int r_ = 0; r_ < rows; r_++) t[sct_off((uint32_t)r_) + lane] = ct[(size_t)(r_ * WAVE + lane) * P.cstride];
      S.sct = t;
    }
  }
  if (c.alive) {
    for (int j = gl; j < Mh; j += G) {
      S.cols[(size_t)gi * mmax + j] = (uint16_t)mi[META_I_COLS + j];
      S.nal[(size_t)gi * mmax + j] = (uint8_t)mi[META_I_COLS + P.max_pos + j];
      S.shift[(size_t)gi * mmax + j] = (uint8_t)(c.bits * (Mh - 1 - j));
    }
    if (!isnan(C_INB(S, gi)))
      for (int i = gl; i < 2 * KT + 5; i += G) S.prior[(size_t)gi * (2 * KT + 5) + i] = mf[meta_f_prior(0) + i];
    if (D.n_intervals == 0 && gl == 0) {
      // cumulative break-count distribution, summed in the reference's order (structural.py:44-49)
      double cacc = 0.0;
      for (int j = 0; j < Mh; j++) {
        cacc += D.break_table[(size_t)Mh * D.max_pos + j];
        S.bdist[gi * mmax + j] = cacc;
      }
    }
  }
  lds_sync();
  const int amax = [&] {
    int v = c.alive ? A : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
    return v;
  }();
  // ---- initial genotype (assemble/mcmc.py:202-208) ----
  {
    GWords<KT> z;
#pragma unroll
    for (int h = 0; h < KT; h++) z.w[h] = 0;
    c.g = z;
  }
  if (c.alive && !resume) {
    LDSP(uint8_t) shift = S.shift + gi * mmax;
    if (U.initial_off >= 0) {
      const int8_t *ini = D.initial + U.initial_off + (size_t)chain * KT * Mh;
#pragma unroll
      for (int h = 0; h < KT; h++) {
        uint64_t x = 0;
        for (int j = 0; j < Mh; j++) x |= (uint64_t)(uint8_t)ini[h * Mh + j] << shift[j];
        set_word<KT>(c.g, h, x);  // no dynamic index into c.g: it must stay in registers (a loop that is not
                                  // unrolled would otherwise put the whole chain context into scratch memory)
      }
    } else {
      const double *dist = mf + meta_f_dist(P.max_ploidy);
      Stream si;
      si.k0 = (uint32_t)D.seed;
      si.k1 = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
      si.c2 = ((uint32_t)chain << 16) | SLOT_INIT;
      si.c3 = (uint32_t)U.stream_id;
      uint64_t n = 0;
#pragma unroll 1
      for (int h = 0; h < KT; h++) {
        uint64_t x = 0;
        for (int j = 0; j < Mh; j++) {
          double s = 0.0;
          for (int a = 0; a < A; a++) s += dist[j * A + a];
          double cacc = 0.0;
          const double uu = stream_double(si, n++);
          int ch = A;
          for (int a = 0; a < A; a++) {
            cacc += dist[j * A + a] / s;
            if (cacc > uu) {
              ch = a;
              break;
            }
          }
          if (ch >= A) ch = A - 1;
          x |= (uint64_t)ch << shift[j];
        }
        set_word<KT>(c.g, h, x);
      }
    }
  }
  if constexpr (PIPE) {
    if (resume) {
      const PipeState *st = reinterpret_cast<const PipeState *>(P.pipe_state) + q;
      if (c.alive) {
        GWords<KT> gr;
#pragma unroll
        for (int h = 0; h < KT; h++) gr.w[h] = st->g[h];
        c.g = gr;
        c.llk = st->llk;
        c.ctr = st->ctr;
        c.mvalid = st->mvalid != 0;
        base = st->step;
        if (gl == 0) {
          S.gval[gi * GV_N + GV_MLO] = st->mlo;
          S.gval[gi * GV_N + GV_MHI] = st->mhi;
        }
        const double *pm = P.pipe_memo + (size_t)q * S.memo_stride;
        for (int i = gl; i < S.memo_stride; i += G) S.memo_tot[gi * S.memo_stride + i] = pm[i];
        if (base >= Sn) c.alive = false;  // finished (or stopped by an error) in an earlier launch
      }
      lds_sync();
    }
  }
  if (!resume) {
    const bool req = c.alive && gl == 0;  // assemble/mcmc.py:303
    const GWords<KT> g0 = c.g;
    if (req) {
#pragma unroll
      for (int h = 0; h < KT; h++) S.pw[h * WAVE + lane] = g0.w[h];
    }
    lds_sync();
    const double v = spec_coop_all<KT, G>(__ballot(req), S.pw, S.shift, S.cols, S.nreads, S.ndict, S.dict, S.gptr, S.bw, S.reuse_on, S.crow, mmax, c.Mh, C_AMASK(c), rpad, lane, nullptr, nullptr, S.bpc, S.bpt, S.sct, S.gbt);
    lds_sync();
    c.llk = __shfl(v, 0, G);
    if (c.alive && gl == 0) {
      for (int t = 0; t < T; t++) {
#pragma unroll
        for (int h = 0; h < KT; h++) S.wst[((size_t)gi * T + t) * KT + h] = g0.w[h];
        S.llk_t[(size_t)gi * T + t] = c.llk;
        S.rngn[(size_t)gi * T + t] = 0;
      }
    }
  }
  lds_sync();
  const double *break_dist = D.break_table + (size_t)Mh * D.max_pos;
  const int n_break_dist = D.n_intervals > 0 ? D.n_intervals : Mh;
  int status = MCHAP_UNIT_OK;

#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
  for (int i_ = 0; i_ < 20; i_++) c.ph[i_] = 0;
  c.pt0 = __builtin_amdgcn_s_memtime();
#endif
  int n_iter = Sn;
  if constexpr (PIPE) {
    int mine = c.alive ? Sn - base : 0;
    if (P.pipe_iters > 0 && mine > P.pipe_iters) mine = P.pipe_iters;
    if (fillonly) mine = 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine = max(mine, __shfl_xor(mine, o, WAVE));
    n_iter = mine;
  }
  // PIPE_EXPORT: a wave with a chain that is not settled yet (it moved after its last full mutation step) keeps
  // stepping, up to n_cap iterations: the coasting kernel would hand that chain straight back
  int n_cap = n_iter;
  if constexpr (PIPE) {
    if ((P.pipe_mode & PIPE_EXPORT) && P.pipe_iters > 0) {
      int mine = c.alive ? Sn - base : 0;
      if (mine > P.pipe_iters_max) mine = P.pipe_iters_max;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mine = max(mine, __shfl_xor(mine, o, WAVE));
      n_cap = max(n_iter, mine);
    }
  }
  int n_done = 0;  // PIPE: compound steps this chain completed in this launch
  for (int it = 0; it < n_iter; it++) {
    const int step = base + it;
    if constexpr (PIPE) {
      if (step >= Sn) c.alive = false;  // this chain is complete; others of the wave go on
    }
    for (int t = TW ? tw : 0; t < (TW ? tw + 1 : T); t++) {
      GPHASE(c, 8);
      if (T > 1) {
        GWords<KT> gt;
#pragma unroll
        for (int h = 0; h < KT; h++) gt.w[h] = S.wst[((size_t)gi * T + t) * KT + h];
        c.g = gt;
        c.llk = S.llk_t[(size_t)gi * T + t];
      }
      c.ctr = (uint64_t)step * STEP_DRAWS;  // the step's draws (philox.hpp)
      c.dcount = 0;  // nothing staged
      const double temp = D.temps[t];
      if (T > 1) {  // the temperature's stream
        lds_sync();
        if (gl == 0) S.gstream[gi * 4 + 2] = ((uint32_t)chain << 16) | (uint32_t)t;
        lds_sync();
      }
      if (c.alive && isnan(c.llk)) {  // assemble/mcmc.py:330-331
        status = MCHAP_UNIT_NAN_LLK;
        c.alive = false;
      }
      spec_mutation<KT, G, false, CTX>(c, S, temp, amax, mmax, nmax, rpad, lane, gi, gl);
#pragma unroll 1
      for (int kind = 0; kind < 3; kind++) {
        if (!spec_structural<KT, G, PIPE, CTX>(c, S, D, kind, temp, break_dist, n_break_dist, mmax, rpad, lane, gi, gl, amax)) {
          status = MCHAP_UNIT_BREAKS;
          c.alive = false;
        }
      }
      if constexpr (TW) {
        // publish this replica's state, then the swap chain: wavefront tt attempts its swap with tt - 1 (tempering.py:61-151)
        // against what tt - 1 holds then; a replica's own state is only touched again by its upper neighbour's later attempt
        {
          const GWords<KT> gp_ = c.g;
          if (lane == 0) {
#pragma unroll
            for (int h = 0; h < KT; h++) xw[(size_t)tw * KT + h] = gp_.w[h];
            xllk[tw] = c.llk;
            if (status != MCHAP_UNIT_OK) *xdead = 1;
          }
        }
        __syncthreads();
        if (*xdead) c.alive = false;  // (the reference raises: every replica of the chain stops; the barriers go on)
        for (int tt = 1; tt < T; tt++) {
          if (tw == tt && c.alive) {
            LDSP(double) pt = S.prior + gi * (2 * KT + 5);
            GWords<KT> gj;
#pragma unroll
            for (int h = 0; h < KT; h++) gj.w[h] = xw[(size_t)(tt - 1) * KT + h];
            const double llk_j = xllk[tt - 1];
            double prior_i = 0.0, prior_j = 0.0;
            if (!isnan(C_INB(S, gi))) {
              prior_i = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(c.g));
              prior_j = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(gj));
            }
            const double ui = c.llk + prior_i, uj = llk_j + prior_j;
            double acc = exp((uj - ui) * temp + (ui - uj) * D.temps[tt - 1]);
            if (acc > 1.0) acc = 1.0;
            const double val = stream_double(ld_stream(S, gi), c.ctr++);
            if (acc >= val) {
              const GWords<KT> gi_ = c.g;
              if (lane == 0) {
#pragma unroll
                for (int h = 0; h < KT; h++) {
                  xw[(size_t)(tt - 1) * KT + h] = gi_.w[h];
                  xw[(size_t)tt * KT + h] = gj.w[h];
                }
                xllk[tt - 1] = c.llk;
                xllk[tt] = llk_j;
              }
            }
          }
          __syncthreads();
        }
        if (c.alive) {  // this replica's state after every attempt that involved it
          GWords<KT> gn;
#pragma unroll
          for (int h = 0; h < KT; h++) gn.w[h] = xw[(size_t)tw * KT + h];
          c.g = gn;
          c.llk = xllk[tw];
          if (gl == 0) {
#pragma unroll
            for (int h = 0; h < KT; h++) S.wst[((size_t)gi * T + t) * KT + h] = gn.w[h];
            S.llk_t[(size_t)gi * T + t] = c.llk;
          }
        }
        __syncthreads();  // (the exchange area is rewritten in the next step)
        lds_sync();
      } else
      if (T > 1) {
        if (c.alive && t > 0) {
          // tempering.py:61-151 with the previous (warmer) temperature
          LDSP(double) pt = S.prior + gi * (2 * KT + 5);
          GWords<KT> gj;
#pragma unroll
          for (int h = 0; h < KT; h++) gj.w[h] = S.wst[((size_t)gi * T + t - 1) * KT + h];
          double llk_j = S.llk_t[(size_t)gi * T + t - 1];
          double prior_i = 0.0, prior_j = 0.0;
          if (!isnan(C_INB(S, gi))) {
            prior_i = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(c.g));
            prior_j = prior_of<KT>(pt, C_INB(S, gi), dosage_words<KT>(gj));
          }
          const double ui = c.llk + prior_i, uj = llk_j + prior_j;
          double acc = exp((uj - ui) * temp + (ui - uj) * D.temps[t - 1]);
          if (acc > 1.0) acc = 1.0;
          const double val = stream_double(ld_stream(S, gi), c.ctr++);
          lds_sync();
          if (acc >= val) {
            const GWords<KT> gi_ = c.g;
            if (gl == 0) {
#pragma unroll
              for (int h = 0; h < KT; h++) S.wst[((size_t)gi * T + t - 1) * KT + h] = gi_.w[h];
              S.llk_t[(size_t)gi * T + t - 1] = c.llk;
            }
            c.g = gj;
            c.llk = llk_j;
          }
        }
        lds_sync();
        if (c.alive && gl == 0) {
          const GWords<KT> gn = c.g;
#pragma unroll
          for (int h = 0; h < KT; h++) S.wst[((size_t)gi * T + t) * KT + h] = gn.w[h];
          S.llk_t[(size_t)gi * T + t] = c.llk;
          S.rngn[(size_t)gi * T + t] = c.ctr;
        }
        lds_sync();
      }
    }
    // record the cold chain (held in registers after the last temperature) in canonical order: SPEC_TB records
    // are collected in LDS and written together, the words as one contiguous run (a full line for K = 4)
    {
      LDSP(uint64_t) tb = S.tbuf + (size_t)gi * SPEC_TB * (KT + 1);
      const int slot = it % SPEC_TB;
      if (c.alive) n_done = it + 1;
      if (c.alive && (!TW || tw == T - 1)) {  // (TW: the cold replica's wavefront records)
        if (gl < KT) {
          const GWords<KT> gr = c.g;
          const uint64_t x = sel_word<KT>(gr, gl);
          int rank = 0;
#pragma unroll
          for (int h = 0; h < KT; h++) rank += (gr.w[h] < x || (gr.w[h] == x && h < gl)) ? 1 : 0;
          tb[slot * KT + rank] = x;
        }
        if (gl == 0) tb[SPEC_TB * KT + slot] = (uint64_t)__double_as_longlong(c.llk);
      }
      // (PIPE: the chains of a wave may be at different steps; one that completes flushes its block then)
      bool last = it + 1 >= n_iter;
      if constexpr (PIPE) {
        if (last && it + 1 < n_cap && wave_any(c.alive && !c.mvalid && step + 1 < Sn)) {
          n_iter = it + 2;
          last = false;
        }
      }
      const bool flush = slot == SPEC_TB - 1 || last || step == Sn - 1;
      if (PIPE ? wave_any(flush && c.alive) : flush) {
        lds_sync();
        if (c.alive && flush && (!TW || tw == T - 1)) {
          const int first = step - slot;  // first step of the block
          const int nw = (slot + 1) * KT;
          uint64_t *tp = reinterpret_cast<uint64_t *>((uintptr_t)S.gptr[gi * GP_N + GP_TRACE]) + (size_t)first * KT;
          for (int i = gl; i < nw; i += G) tp[i] = tb[i];
          if (gl <= slot)
            reinterpret_cast<uint64_t *>((uintptr_t)S.gptr[gi * GP_N + GP_LLK])[first + gl] = tb[SPEC_TB * KT + gl];
        }
        lds_sync();
      }
    }
  }
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
  if (threadIdx.x == 0)
    for (int i_ = 0; i_ < 20; i_++) atomicAdd(&g_stats[i_ < 12 ? 24 + i_ : 36 + i_], c.ph[i_]);
  for (int i_ = 0; i_ < 20; i_++) c.ph[i_] = 0;
  c.pt0 = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (PIPE) {
    if (P.pipe_mode & (PIPE_EXPORT | PIPE_FILLONLY)) {
      // The table describes genotype generation memo_gen and is wiped lazily, when a structural step next looks at it.
      // A chain that moved in the last structural step of this launch still carries the table of its previous genotype:
      // it must not leave the kernel like that -- whoever loads it (the resuming launch, the table completion, the
      // coasting kernel) takes every entry it finds for the current genotype's.
      if (S.memo_stride != 0 && wave_any(c.alive && c.gen != c.memo_gen)) {
        if (c.alive && c.gen != c.memo_gen) {
          LDSP(double) all = S.memo_tot + gi * S.memo_stride;
          for (int i = gl; i < S.memo_stride; i += G) all[i] = NAN;
          c.memo_gen = c.gen;
        }
        lds_sync();
      }
      // the interval memo of the current genotype, completed (both step types) -- unless a PIPE_FILLONLY launch will
      // do that with several wavefronts per chain -- then the hand-over record
      if ((fillonly || parts_eff <= 1) && !(P.flags & 32) && !(P.pipe_mode & PIPE_NOFILL)) {
#pragma unroll 1
        for (int kind = 3; kind < 5; kind++)
          spec_structural<KT, G, true>(c, S, D, kind, D.temps[0], break_dist, n_break_dist, mmax, rpad, lane, gi, gl);
      }
      lds_sync();
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
      if (threadIdx.x == 0)
        for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_stats[36 + i_], c.ph[i_]);
#endif
      if (fillonly) {
        if (listed && c.alive) {  // this group's share of the entries (known ones are rewritten with their value)
          double *pm = P.pipe_memo + (size_t)q * S.memo_stride;
          for (int i = gl; i < S.memo_stride; i += G)
            if (i % parts_eff == my_part) pm[i] = S.memo_tot[gi * S.memo_stride + i];
        }
      } else if (listed) {
        PipeState *st = reinterpret_cast<PipeState *>(P.pipe_state) + q;
        const GWords<KT> ge = c.g;
        if (gl == 0) {
#pragma unroll
          for (int h = 0; h < KT; h++) st->g[h] = ge.w[h];
          st->llk = c.llk;
          st->ctr = c.ctr;
          st->mlo = S.gval[gi * GV_N + GV_MLO];
          st->mhi = S.gval[gi * GV_N + GV_MHI];
          st->mvalid = (c.alive && c.mvalid && c.memo_on) ? 1 : 0;
          st->step = c.alive ? base + n_done : Sn;
        }
        if (c.alive) {
          double *pm = P.pipe_memo + (size_t)q * S.memo_stride;
          for (int i = gl; i < S.memo_stride; i += G) pm[i] = S.memo_tot[gi * S.memo_stride + i];
        }
      }
    }
  }
  if (status != MCHAP_UNIT_OK && gl == 0) atomicMax(&D.status[u], status);
}

}  // namespace mchap
