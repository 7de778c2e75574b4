// De-novo MCMC sampler, "steady-state" form for MI355X (gfx950): L lanes per chain in the common case, the whole
// wavefront for one chain when a chain needs likelihoods.
//
// Observation (tools/steps_sweep.py).  On well-covered loci a chain reaches its mode within a few steps
// and then never moves: every later compound step (assemble/mutation.py:164-246, assemble/structural.py:590-673) only
// has to establish that its uniforms fall outside the move probabilities of the current genotype.  Those probabilities
// are a property of the genotype, not of the step, so they are remembered as integer thresholds on the 53-bit uniform:
//   * mutation step: LO <= u53 < HI for each of its K*M uniforms  <=>  no sub-step moves, whatever the shuffle;
//   * interval step (type, start, stop): u53 >= T  <=>  the step's categorical draw lands on "stay".
// With the thresholds known a compound step is a handful of integer compares on Philox words -- scalar work per chain.
// A group of 16 lanes per chain (denovo_spec_kernel.hpp) spends 16 lanes on that scalar work; here a chain owns only
// L = 1/2/4/8/16 lanes (they share the Philox blocks of the step and replicate the scalar logic), so a wavefront
// advances 64 / L chains per instruction stream.
//
// Whenever a chain cannot decide a compound step from its thresholds (they are unknown for the genotype, or a uniform
// falls into a move region), the WHOLE wavefront serves that chain:
//   * mutation step: spec_mutation<K, 64> of the speculative kernel (all sub-steps at once, one per lane);
//   * interval steps: serve_structural() below -- the interval the chain is waiting for plus as many of the chain's
//     still unknown intervals as fit are enumerated together, one option per lane (up to 64 options per round), their
//     likelihoods probed / evaluated, and the thresholds of all of them stored.  Chains with unknown thresholds are
//     also served between steps while their genotype is stable, so the tables are complete a few steps after a chain
//     has settled and the rest of the run never leaves the integer fast path.
// Served values are what the sequential algorithm computes (same factors, same order), decisions use the same float64
// compares, and the thresholds are exact (mutation) or conservative (interval steps: 27-bit, a uniform within 2^-27 of
// the threshold is decided by a full evaluation), so the traces are bit-identical to the other kernels' and to the
// oracle's.  Single temperature only (parallel tempering runs on the speculative kernel).
#pragma once
#include "denovo_spec_kernel.hpp"

namespace mchap {

constexpr uint32_t MEMO_UNKNOWN = 0xFFFFFFFFu;  // interval step not evaluated for the current genotype
constexpr uint32_t MEMO_NOOPT = 0xFFFFFFFEu;    // the step has no options: it consumes no draw
constexpr int LANE_TB = 4;                      // trace records buffered per chain

__host__ __device__ inline int lane_window(int Mmax) {  // staged draws per chain for the structural steps
  const int w = 3 * Mmax + 2;                           // the most one compound step can consume
  return w < 32 ? 32 : ((w + 1) & ~1);
}
__host__ __device__ inline int lane_extra(int Mmax) {  // draws staged behind a mutation step's uniforms
  const int e = 2 * Mmax;
  const int lim = lane_window(Mmax);
  return (e < 16 ? 16 : (e > lim ? lim : e)) & ~1;
}

struct LaneLds {
  // per chain
  LDSP(uint64_t) win;      // [NC][WIN]
  LDSP(uint32_t) memo;     // [NC][2][tri]
  LDSP(uint64_t) bcum;     // [NC][Mmax]  break count: first j with u53 < bcum[j]
  LDSP(uint64_t) tbuf;     // [NC][LANE_TB][K + 1]
  LDSP(uint64_t) gptr;     // [NC][GP_N]
  LDSP(double) gval;       // [NC][GV_N]
  LDSP(uint32_t) gstream;  // [NC][4]
  LDSP(double) prior;      // [NC][2K+5]
  LDSP(uint16_t) cols;     // [NC][Mmax]
  LDSP(uint8_t) shift;     // [NC][Mmax]
  LDSP(uint8_t) nal;       // [NC][Mmax]
  LDSP(uint8_t) ord;       // [NC][Mmax + 1]
  LDSP(uint16_t) nreads;   // [NC]
  LDSP(uint16_t) ndict;    // [NC]
  LDSP(uint8_t) tct;       // [NC][tab_bytes] the unit's coded table (copied once: the evaluations then never leave the CU)
  LDSP(double) tcw;        // [NC][rpad]      ... and its read weights
  LDSP(double) dict;       // [NC][DICT_MAX]  ... and its dictionary
  int tab_bytes;
  // per wave: the serving context
  LDSP(uint64_t) pw;       // [K][64]
  LDSP(double) ptab;       // [64]
  LDSP(double) optp;       // [nopt][64]
  LDSP(double) optl;       // [nopt][64]
  LDSP(double) ln;         // [SPEC_LN]
  LDSP(double) lninv;      // [SPEC_LN]
  LDSP(uint16_t) permtab;  // [nmax]
  LDSP(uint8_t) ktab;      // [nmax]
  LDSP(uint64_t) draws;    // [ndraws]
  LDSP(uint64_t) bw;       // [K]
  LDSP(int) cum;           // [64]
  LDSP(uint16_t) itab;     // [tri] start | stop << 8 of triangular index
  int win_n, tri, ndraws;
};

__host__ __device__ inline size_t lane_lds_bytes(int K, int Mmax, int Amax, int L, int tab_bytes, int rpad) {
  const int NC = 64 / L;
  const int tri = spec_memo_entries(Mmax);
  const int nmax = K * Mmax;
  const int nopt = Amax > 1 ? Amax - 1 : 1;
  size_t b = 0;
  b += (size_t)8 * NC * lane_window(Mmax);
  b += (size_t)8 * NC * Mmax;
  b += (size_t)8 * NC * LANE_TB * (K + 1);
  b += (size_t)8 * NC * GP_N;
  b += (size_t)8 * NC * GV_N;
  b += (size_t)8 * NC * (2 * K + 5);
  b += (size_t)16 * NC;                 // gstream
  b += (size_t)4 * NC * 2 * tri;        // memo
  b += (size_t)2 * NC * Mmax;           // cols
  b += (size_t)NC * Mmax * 2;           // shift, nal
  b += (size_t)NC * (Mmax + 1);         // ord
  b = (b + 1) & ~(size_t)1;
  b += (size_t)2 * NC * 2;              // nreads, ndict
  b = (b + 15) & ~(size_t)15;
  b += (size_t)8 * K * 64;              // pw
  b += (size_t)8 * 64;                  // ptab
  b += (size_t)8 * nopt * 64 * 2;       // optp, optl
  b += (size_t)8 * SPEC_LN * 2;
  b += (size_t)8 * spec_draws(K, Mmax); // draws
  b += (size_t)NC * ((size_t)tab_bytes + (size_t)8 * rpad + (size_t)8 * DICT_MAX);  // unit tables
  b += (size_t)8 * K;                   // bw
  b += (size_t)4 * 64;                  // cum
  b += (size_t)2 * nmax;                // permtab
  b += (size_t)2 * tri;                 // itab
  b += (size_t)nmax;                    // ktab
  return (b + 63) & ~(size_t)63;
}

__device__ __forceinline__ uint64_t u53_of(uint64_t w) {
  return ((uint64_t)((uint32_t)w >> 5) << 26) | (uint64_t)((uint32_t)(w >> 32) >> 6);
}
__device__ __forceinline__ Stream ld_stream_l(LDSP(uint32_t) gstream, int ci) {
  LDSP(uint32_t) w = gstream + ci * 4;
  Stream st;
  st.k0 = w[0];
  st.k1 = w[1];
  st.c2 = w[2];
  st.c3 = w[3];
  return st;
}
// draws base .. base + count - 1 of a stream into tab[0 .. count), Philox blocks dealt round-robin to the chain's L lanes
__device__ __forceinline__ void stage_draws_l(const Stream &s, uint64_t base, int count, LDSP(uint64_t) tab, int sl, int L) {
  const uint64_t b0 = base >> 1;
  const int nblk = (int)(((base + (uint64_t)count + 1) >> 1) - b0);
  for (int b = sl; b < nblk; b += L) {
    uint32_t o[4];
    const uint64_t blk = b0 + (uint64_t)b;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), s.c2, s.c3, s.k0, s.k1, o);
    const long long i0 = (long long)(blk << 1) - (long long)base;
    if (i0 >= 0 && i0 < count) tab[i0] = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
    if (i0 + 1 >= 0 && i0 + 1 < count) tab[i0 + 1] = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
  }
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, WAVE);
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, WAVE);
  return ((uint64_t)hi << 32) | lo;
}
// smallest integer t with t * 2^-53 >= x (x in [0, 1]; clamped), i.e. u >= x  <=>  u53 >= t, and u < x  <=>  u53 < t
__device__ __forceinline__ uint64_t ceil53(double x) {
  if (!(x > 0.0)) return 0ull;
  if (x >= 1.0) return 1ull << 53;
  return (uint64_t)ceil(x * 9007199254740992.0);
}
__device__ __forceinline__ uint32_t memo_threshold(double tot) {  // no move <= (a >> 5) >= threshold
  return (uint32_t)((ceil53(tot) + ((1ull << 26) - 1ull)) >> 26);
}

// Per-chain state held (replicated) in the registers of the chain's L lanes.
template <int KT>
struct LChain {
  GWords<KT> g;
  double llk;
  uint64_t ctr;        // next draw of the chain's stream
  uint64_t lo, hi;     // mutation step: no sub-step moves iff lo <= u53 < hi for every uniform (valid iff mvalid)
  int doff, dcount;    // window: entry doff holds draw ctr
  int Mh, bits;
  int n_unknown;       // interval-step thresholds still unknown for the current genotype
  int cursor;          // where the next fill round looks for unknown entries
  bool alive, mvalid, stable;
};

// The chain `cs` of the wave as a wave-uniform speculative-kernel context + its LDS tables as "group 0".
template <int KT>
__device__ __forceinline__ void lane_context(const LaneLds &LL, int cs, int mmax, SpecLds &S) {
  S.pw = LL.pw;
  S.ptab = LL.ptab;
  S.optp = LL.optp;
  S.optl = LL.optl;
  S.ln = LL.ln;
  S.lninv = LL.lninv;
  S.permtab = LL.permtab;
  S.ktab = LL.ktab;
  S.draws = LL.draws;
  S.ndraws = LL.ndraws;
  S.dict = LL.dict + (size_t)cs * DICT_MAX;
  S.bw = LL.bw;
  S.prior = LL.prior + cs * (2 * KT + 5);
  S.cols = LL.cols + cs * mmax;
  S.shift = LL.shift + cs * mmax;
  S.nal = LL.nal + cs * mmax;
  S.nreads = LL.nreads + cs;
  S.ndict = LL.ndict + cs;
  S.gptr = LL.gptr + cs * GP_N;
  S.gval = LL.gval + cs * GV_N;
  S.gstream = LL.gstream + cs * 4;
  S.memo_stride = 0;
  S.memo_tot = nullptr;
}
template <int KT>
__device__ __forceinline__ void lane_context_tabs(const LaneLds &LL, int cs, int rpad, SpecLds &S) {
  S.lds_ct = (LDSP(const uint8_t))(LL.tct + (size_t)cs * LL.tab_bytes);
  S.lds_cw = (LDSP(const double))(LL.tcw + (size_t)cs * rpad);
  S.bpc = nullptr;
  S.bpt = nullptr;
}

template <int KT>
__device__ __forceinline__ GWords<KT> bcast_words(const GWords<KT> g, int src) {
  GWords<KT> r;
#pragma unroll
  for (int h = 0; h < KT; h++) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)g.w[h], src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(g.w[h] >> 32), src);
    r.w[h] = ((uint64_t)hi << 32) | lo;
  }
  return r;
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double bcast_f64(double v, int src) {
  return __longlong_as_double((long long)bcast_u64((uint64_t)__double_as_longlong(v), src));
}

// One serving round for chain `cs` (wave-uniform): the interval step the chain waits for (has_req: type, start, stop, its
// uniform in `uw`) plus unknown interval steps of the same genotype, one option per lane.  Returns in every lane:
// filled = thresholds stored, moved / nopt_req for the required step; the new genotype and llk in cu when moved.
template <int KT>
struct ServeResult {
  int filled;
  bool moved;
  int n_opt_req;
};

template <int KT>
__device__ __forceinline__ ServeResult<KT> serve_structural(Grp<KT> &cu, const SpecLds &S, const LaneLds &LL, int cs, bool has_req,
                                                            int req_type, int req_idx, uint64_t uw, int cursor, int mmax, int rpad,
                                                            int lane) {
  ServeResult<KT> R;
  R.filled = 0;
  R.moved = false;
  R.n_opt_req = 0;
  const int Mh = cu.Mh;
  const int tri_h = spec_memo_entries(Mh);  // entries with stop <= Mh come first in the triangular order
  LDSP(uint32_t) memo = LL.memo + (size_t)cs * 2 * LL.tri;
  LDSP(double) pt = S.prior;
  const uint64_t full = mask_of(cu.bits, Mh, 0, Mh);
  // ---- candidate tasks, one per lane: lane 0 the required step, the others scan 2 * tri_h entries from the cursor ----
  int my_type = 0, my_idx = 0;
  bool cand = false;
  if (lane == 0 && has_req) {
    my_type = req_type;
    my_idx = req_idx;
    cand = true;
  } else {
    const int total = 2 * tri_h;
    const int t = lane - (has_req ? 1 : 0);
    if (t < total) {
      int e = cursor + t;
      if (e >= total) e -= total;
      my_type = e >= tri_h ? 1 : 0;
      my_idx = e - my_type * tri_h;
      cand = memo[my_type * LL.tri + my_idx] == MEMO_UNKNOWN && !(has_req && my_type == req_type && my_idx == req_idx);
    }
  }
  uint32_t lin = 0, lout = 0;
  int n_opt = 0;
  uint64_t msk = 0;
  bool was_noopt = false;
  if (cand) {
    const uint32_t se = LL.itab[my_idx];
    msk = mask_of(cu.bits, Mh, (int)(se & 255u), (int)(se >> 8));
    lin = seg_labels<KT>(cu.g, msk);
    lout = seg_labels<KT>(cu.g, full & ~msk);
    n_opt = my_type == 0 ? recombination_n_options(lin, lout, KT) : dosage_n_options(lin, lout, KT);
    if (n_opt == 0 && !(lane == 0 && has_req)) {
      memo[my_type * LL.tri + my_idx] = MEMO_NOOPT;  // nothing to evaluate
      cand = false;
      was_noopt = true;
    }
  }
  R.filled += __popcll(__ballot(was_noopt));  // entries resolved as "no options" are resolved
  if (lane == 0 && has_req) R.n_opt_req = n_opt;
  R.n_opt_req = __builtin_amdgcn_readfirstlane(R.n_opt_req);
  if (has_req && R.n_opt_req == 0) {
    // the step the chain waits for has no options (no draw): remember, and use the round for the others
    if (lane == 0) memo[req_type * LL.tri + req_idx] = MEMO_NOOPT;
    if (lane == 0) cand = false;
    R.filled += 1;
  }
  // ---- slots: tasks in lane order while their options fit into 64 ----
  int cum = cand ? n_opt : 0;
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    const int v = __shfl_up(cum, o, WAVE);
    if (lane >= o) cum += v;
  }
  const bool sel = cand && cum <= WAVE;
  LL.cum[lane] = cum <= WAVE ? cum : 0x7fffffff;  // inclusive sums: monotone, unselected tail = +inf
  lds_sync();
  const int n_slots = [&] {
    const unsigned long long m = __ballot(sel);
    if (!m) return 0;
    const int last = 63 - __clzll((long long)m);
    return __builtin_amdgcn_readlane(cum, last);
  }();
  // slot `lane`: its task = first lane t with cum[t] > lane
  int task = 0;
  {
    int lo = 0, hi = WAVE - 1;
#pragma unroll
    for (int it = 0; it < 6; it++) {
      const int mid = (lo + hi) >> 1;
      if (LL.cum[mid] > lane) hi = mid;
      else lo = mid + 1;
    }
    task = lo;
  }
  const bool prop = lane < n_slots;
  const int t_type = __shfl(my_type, task, WAVE);
  const int t_nopt = __shfl(n_opt, task, WAVE);
  const int t_base = __shfl(cum, task, WAVE) - t_nopt;
  const uint32_t t_lin = (uint32_t)__shfl((int)lin, task, WAVE);
  const uint32_t t_lout = (uint32_t)__shfl((int)lout, task, WAVE);
  const uint64_t t_msk = shfl_u64(msk, task);
  const int my_o = lane - t_base;
  // ---- my option: the my_o-th in the reference's enumeration order (structural.py:121-178 / 240-307) ----
  const GWords<KT> cg = cu.g;
  GWords<KT> pw = cg;
  uint32_t oin = 0;
  if (prop) {
    const uint32_t hd = dosage_of_labels(t_lin, t_lout, KT, true);
    int cnt = 0;
    if (t_type == 0) {
#pragma unroll
      for (int h0 = 0; h0 < KT; h0++) {
#pragma unroll
        for (int h1 = h0 + 1; h1 < KT; h1++) {
          const bool valid = nib(hd, h0) != 0 && nib(hd, h1) != 0 && nib(t_lin, h0) != nib(t_lin, h1) && nib(t_lout, h0) != nib(t_lout, h1);
          if (valid) {
            if (cnt == my_o) {
              uint32_t o = nib_set(t_lin, h0, nib(t_lin, h1));
              oin = nib_set(o, h1, nib(t_lin, h0));
            }
            cnt++;
          }
        }
      }
    } else {
      const uint32_t sd = dosage_of_labels(t_lin, t_lout, KT, false);
#pragma unroll
      for (int h0 = 0; h0 < KT; h0++) {
#pragma unroll
        for (int h1 = 0; h1 < KT; h1++) {
          const bool valid = nib(hd, h0) != 0 && nib(sd, h0) != 1 && nib(sd, h1) != 0 && nib(t_lin, h0) != nib(t_lin, h1);
          if (valid) {
            if (cnt == my_o) oin = nib_set(t_lin, h0, nib(t_lin, h1));
            cnt++;
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < KT; h++) pw.w[h] = (cg.w[h] & ~t_msk) | (sel_word<KT>(cg, (int)nib(oin, h)) & t_msk);
  }
  const double llk_i = spec_eval<KT, 64, true>(prop, pw, cu, S, mmax, rpad, lane);
  if (prop) {
    double lprior_ratio = 0.0;
    if (!isnan(C_INB(S, 0)))
      lprior_ratio = prior_of<KT>(pt, C_INB(S, 0), dosage_of_labels(oin, t_lout, KT, true)) - prior_of<KT>(pt, C_INB(S, 0), dosage_words<KT>(cg));
    const int n_return = t_type == 0 ? recombination_n_options(oin, t_lout, KT) : dosage_n_options(oin, t_lout, KT);
    const double lproposal_ratio = S.lninv[n_return] - S.lninv[t_nopt];
    const double mh = ((llk_i - cu.llk) + lprior_ratio) + lproposal_ratio;  // temperature 1: x * 1.0 == x
    S.ptab[lane] = exp(fmin(0.0, mh) - S.ln[t_nopt]);
  }
  lds_sync();
  // ---- per task: cumulative option probabilities in order; the required step's categorical draw ----
  int choice = -1;
  if (sel) {
    const int base = cum - n_opt;
    double cacc = 0.0;
    if (lane == 0 && has_req) {
      const double u = draw_double(uw);
      for (int o = 0; o < n_opt; o++) {
        cacc += S.ptab[base + o];
        if (cacc > u) {
          choice = base + o;
          break;
        }
      }
      if (choice < 0) memo[my_type * LL.tri + my_idx] = memo_threshold(cacc);
    } else {
      for (int o = 0; o < n_opt; o++) cacc += S.ptab[base + o];
      memo[my_type * LL.tri + my_idx] = memo_threshold(cacc);
    }
  }
  R.filled += __popcll(__ballot(sel && choice < 0));
  const int ch = __builtin_amdgcn_readfirstlane(choice);  // lane 0 holds the required task when there is one
  if (has_req && R.n_opt_req > 0 && ch >= 0) {
    R.moved = true;
    cu.g = bcast_words<KT>(pw, ch);
    cu.llk = bcast_f64(llk_i, ch);
  }
  lds_sync();
  return R;
}

// ---------------------------------------------------------------------------------------------------------
// Chain state handed between the two kernels of the pipeline (one record per chain in the workspace)
// ---------------------------------------------------------------------------------------------------------
struct LaneState {
  uint64_t g[MCHAP_MAX_PLOIDY];  // haplotype words
  double llk;
  uint64_t ctr;                  // next draw: the first draw of compound step `phase` of step `step`
  uint64_t lo, hi;               // mutation bounds (valid iff flags & LS_MVALID)
  int32_t step;                  // next step to run; == steps when the chain is finished (or dead)
  int32_t phase;                 // compound step to resume at: 0 mutation, 1..3 the structural steps
  int32_t flags;
  int32_t pad;
};
static_assert(sizeof(LaneState) == 112, "LaneState layout");
enum { LS_MVALID = 1, LS_SETTLED = 2 };  // SETTLED: every threshold of the genotype is known, the steady kernel may run it
enum { LANE_MODE_RESUME = 1, LANE_MODE_PARK = 2 };

// profiling builds (make stats / make phases): wave clock per region and event counts, summed over the launch
#if defined(MCHAP_PHASES)
#define LPH(i)                                                   \
  do {                                                           \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    lph[i] += t_ - lpt0;                                         \
    lpt0 = t_;                                                   \
  } while (0)
#define LCNT(i, n) lph[i] += (unsigned long long)(n)
#else
#define LPH(i)
#define LCNT(i, n)
#endif

// ---------------------------------------------------------------------------------------------------------
// The integer fast path, shared by both kernels
// ---------------------------------------------------------------------------------------------------------
// Mutation compound step of a chain whose bounds are valid: this lane's share of the Philox blocks behind draw `base`
// (the step's first uniform): uniforms 0 .. n-1 are tested in registers, the E draws behind them go to the chain's window.
// Returns false if one of this lane's uniforms falls outside [lo, hi).
__device__ __forceinline__ bool fast_mutation(const Stream &st, uint64_t base, int n, int E, uint64_t lo, uint64_t hi, LDSP(uint64_t) win,
                                              int sl, int L) {
  bool ok = true;
  const uint64_t b0 = base >> 1;
  const int nblk = (int)(((base + (uint64_t)(n + E) + 1) >> 1) - b0);
  for (int b = sl; b < nblk; b += L) {
    uint32_t o[4];
    const uint64_t blk = b0 + (uint64_t)b;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), st.c2, st.c3, st.k0, st.k1, o);
    const int i0 = (int)((long long)(blk << 1) - (long long)base);  // index of the block's first draw: >= -1
    const uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32), w1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
    const uint64_t x0 = u53_of(w0), x1 = u53_of(w1);
    const bool in0 = x0 >= lo && x0 < hi, in1 = x1 >= lo && x1 < hi;
    if (i0 >= 0 && i0 < n) ok = ok && in0;
    if (i0 + 1 < n) ok = ok && in1;
    if (i0 >= n && i0 < n + E) win[i0 - n] = w0;
    if (i0 + 1 >= n && i0 + 1 < n + E) win[i0 + 1 - n] = w1;
  }
  return ok;
}

// One structural compound step on thresholds alone.  The chain's window holds draws from its current one on (entry
// `doff`, `dcount` valid entries).  Outcome: done (nothing moves: doff_end = window offset behind the step), or exact
// (the thresholds cannot decide: zeros / n_int / doff1 describe the drawn intervals), or bad (structural.py:49-50), or
// short_ (the window ran out: nothing may be concluded; refill and call again).
struct SFast {
  bool done, exact, bad, short_;
  uint64_t zeros;
  int n_int, doff1, doff_end;
};
__device__ __forceinline__ SFast fast_structural(int kind, uint64_t pthr, int n_intervals_fixed, int Mh, LDSP(uint64_t) win, int doff0, int dcount,
                                                 LDSP(uint64_t) bc, LDSP(uint32_t) memo /* of the step's type */, bool memo_iv) {
  SFast R;
  R.done = R.exact = R.bad = R.short_ = false;
  R.zeros = 0;
  R.n_int = 0;
  int doff = doff0;
  auto fetch = [&](int i) -> uint64_t {
    if (i < dcount) return win[i];
    R.short_ = true;
    return 0ull;
  };
  bool doit = u53_of(fetch(doff)) < pthr;  // rand() <= p
  doff++;
  if (doit && kind < 2) {
    int nb;
    if (n_intervals_fixed > 0) {
      doff++;  // break_dist = [0, ..., 0, 1]: the draw is consumed (assemble/mcmc.py:214-217)
      nb = n_intervals_fixed - 1;
    } else {
      const uint64_t x = u53_of(fetch(doff));
      doff++;
      nb = Mh;
      for (int j = 0; j < Mh; j++)
        if (x < bc[j]) {
          nb = j;
          break;
        }
    }
    if (nb >= Mh) {
      R.bad = true;
      doit = false;
    } else {
      uint64_t ind = 0;
      for (int i = 1; i < Mh; i++) ind |= 1ull << i;
      for (int b = 0; b < nb; b++) {
        const int no = __popcll(ind);
        if (no == 0) break;
        int k = 0;
        if (no > 1) {
          k = (int)__umulhi((uint32_t)fetch(doff), (uint32_t)no);
          doff++;
        }
        uint64_t t = ind;
        while (k-- > 0) t &= t - 1;
        ind &= ~(t & (~t + 1));
      }
      R.zeros = ~ind & ((1ull << (Mh + 1)) - 1ull);
      R.n_int = nb + 1;
    }
  } else if (doit) {
    R.zeros = 1ull | (1ull << Mh);
    R.n_int = 1;
  }
  R.doff1 = doff;
  if (!doit) {
    R.done = !R.bad;
  } else {
    // order-free check: every interval known, and every consumed uniform beyond the largest threshold
    bool unknown = !memo_iv;
    int n_cons = 0;
    uint32_t mx = 0;
    uint64_t z = R.zeros;
    for (int qq = 0; qq < R.n_int; qq++) {
      const int start = __ffsll((long long)z) - 1;
      z &= z - 1;
      const int stop = __ffsll((long long)z) - 1;
      const uint32_t t = memo[spec_memo_index(start, stop)];
      if (t == MEMO_UNKNOWN) unknown = true;
      else if (t != MEMO_NOOPT) {
        n_cons++;
        mx = t > mx ? t : mx;
      }
    }
    if (!unknown) {
      bool low = false;
      for (int k = 0; k < n_cons; k++) low = low || (((uint32_t)fetch(doff + (R.n_int - 1) + k)) >> 5) < mx;
      if (!low) {
        doff += R.n_int - 1 + n_cons;
        R.done = true;
      }
    }
    R.exact = !R.done;
  }
  R.doff_end = doff;
  if (R.short_) R.done = R.exact = R.bad = false;
  return R;
}
// The same step with few dependent LDS round trips (the common shapes: at most 3 breaks, at least 5 positions, at most
// FS_PRE draws): the first draws of the window and the first break thresholds are read at static offsets in one batch,
// the drawn intervals' thresholds in a second one, everything else is register arithmetic.  Falls back to
// fast_structural() otherwise; identical outcomes.
constexpr int FS_PRE = 12;
__device__ __forceinline__ SFast fast_structural_ll(int kind, uint64_t pthr, int n_intervals_fixed, int Mh, LDSP(uint64_t) win, int doff0,
                                                    int dcount, LDSP(uint64_t) bc, LDSP(uint32_t) memo, bool memo_iv) {
  const int avail = dcount - doff0;
  if (!memo_iv || n_intervals_fixed > 0 || Mh < 5) return fast_structural(kind, pthr, n_intervals_fixed, Mh, win, doff0, dcount, bc, memo, memo_iv);
  SFast R;
  R.done = R.exact = R.bad = R.short_ = false;
  R.zeros = 0;
  R.n_int = 0;
  R.doff1 = doff0;
  R.doff_end = doff0;
  LDSP(uint64_t) wb = win + doff0;
  if (kind == 2) {
    // whole-haplotype dosage step: decision, then one interval [0, Mh)
    if (avail < 2) {
      R.short_ = true;
      return R;
    }
    const uint64_t w0 = wb[0];
    const uint32_t a1 = (uint32_t)wb[1];
    const uint32_t t = memo[spec_memo_index(0, Mh)];
    const bool doit = u53_of(w0) < pthr;
    R.doff1 = doff0 + 1;
    if (!doit) {
      R.done = true;
      R.doff_end = doff0 + 1;
      return R;
    }
    R.zeros = 1ull | (1ull << Mh);
    R.n_int = 1;
    if (t == MEMO_NOOPT) {
      R.done = true;
      R.doff_end = doff0 + 1;
    } else if (t != MEMO_UNKNOWN && (a1 >> 5) >= t) {
      R.done = true;
      R.doff_end = doff0 + 2;
    } else {
      R.exact = true;
      R.doff_end = doff0 + 1;
    }
    return R;
  }
  // ---- batch 1: draws 0, 1 in full, the first words of draws 2 .. FS_PRE-1, break thresholds 0 .. 3 ----
  const int nv = avail < FS_PRE ? avail : FS_PRE;
  if (nv < 2) {
    R.short_ = true;
    return R;
  }
  const uint64_t w0 = wb[0], w1 = wb[1];
  uint32_t a[FS_PRE];
#pragma unroll
  for (int i = 2; i < FS_PRE; i++) a[i] = i < nv ? (uint32_t)wb[i] : 0u;
  const uint64_t b0 = bc[0], b1 = bc[1], b2 = bc[2], b3 = bc[3];  // Mh >= 5
  const bool doit = u53_of(w0) < pthr;
  if (!doit) {
    R.done = true;
    R.doff1 = R.doff_end = doff0 + 1;
    return R;
  }
  const uint64_t x = u53_of(w1);
  const int nb = x < b0 ? 0 : (x < b1 ? 1 : (x < b2 ? 2 : (x < b3 ? 3 : 4)));
  if (nb > 3) return fast_structural(kind, pthr, n_intervals_fixed, Mh, win, doff0, dcount, bc, memo, memo_iv);
  // break points: b-th draw picks among the Mh - 1 - b remaining interior points (all > 1 here: Mh >= 5, b <= 2)
  uint64_t ind = ((1ull << Mh) - 1ull) & ~1ull;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    if (b < nb) {
      int k = (int)__umulhi(a[2 + b], (uint32_t)(Mh - 1 - b));
      uint64_t t = ind;
      while (k-- > 0) t &= t - 1;
      ind &= ~(t & (~t + 1));
    }
  }
  const int used = 2 + nb;  // decision, break count, nb break points
  R.doff1 = doff0 + used;
  R.zeros = ~ind & ((1ull << (Mh + 1)) - 1ull);
  R.n_int = nb + 1;
  // ---- batch 2: thresholds of the (up to 4) intervals ----
  uint32_t th[4];
  {
    uint64_t z = R.zeros;
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      th[qq] = MEMO_NOOPT;
      if (qq <= nb) {
        const int start = __ffsll((long long)z) - 1;
        z &= z - 1;
        const int stop = __ffsll((long long)z) - 1;
        th[qq] = memo[spec_memo_index(start, stop)];
      }
    }
  }
  bool unknown = false;
  int n_cons = 0;
  uint32_t mx = 0;
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    if (th[qq] == MEMO_UNKNOWN) unknown = true;
    else if (th[qq] != MEMO_NOOPT) {
      n_cons++;
      mx = th[qq] > mx ? th[qq] : mx;
    }
  }
  const int s = used + nb;  // first uniform: behind the nb shuffle draws
  const int last = s + n_cons;  // one past the last draw the step consumes when nothing moves
  if (last > nv) {  // beyond the batch: the window's end (refill) or just this routine's reach
    if (last > avail || used > avail) {
      R.short_ = true;
      R.zeros = 0;
      R.n_int = 0;
      return R;
    }
    return fast_structural(kind, pthr, n_intervals_fixed, Mh, win, doff0, dcount, bc, memo, memo_iv);
  }
  if (!unknown) {
    bool low = false;
#pragma unroll
    for (int i = 2; i < FS_PRE; i++) low = low || (i >= s && i < last && (a[i] >> 5) < mx);
    if (!low) {
      R.done = true;
      R.doff_end = doff0 + last;
      return R;
    }
  }
  R.exact = true;
  R.doff_end = R.doff1;
  return R;
}

__device__ __forceinline__ uint64_t decision_threshold(double pstep) {  // rand() <= p  <=>  u53 < threshold
  return pstep < 0.0 ? 0ull : (pstep >= 1.0 ? (1ull << 53) : (uint64_t)floor(pstep * 9007199254740992.0) + 1ull);
}

// ---------------------------------------------------------------------------------------------------------
// Settling kernel: runs chains with the serving code at hand until their thresholds are complete (then parks them
// for the steady kernel), or to the end.
// ---------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(64, 1) void denovo_settle_kernel(const SimtParams P, const int lsh, const int mode) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int L = 1 << lsh;
  const int NC = WAVE >> lsh;
  const int lane = threadIdx.x;
  const int ci = lane >> lsh, sl = lane & (L - 1);
  const int Cn = D.chains, Sn = D.steps;
  const int mmax = P.max_pos, nmax = KT * P.max_pos;
  const int rpad = D.rpad;
  const bool resume = (mode & LANE_MODE_RESUME) != 0, may_park = (mode & LANE_MODE_PARK) != 0;
  const long long q = (long long)blockIdx.x * NC + ci;  // chain index
  const long long n_chains = (long long)P.n_units * Cn;
  LaneState *state = reinterpret_cast<LaneState *>(P.lane_state);
  if (resume) {
    // nothing to do for this wave?  (finished chains, and settled ones the steady kernel will pick up)
    bool want = false;
    if (q < n_chains) {
      const int st_step = state[q].step, st_flags = state[q].flags;
      want = st_step < Sn && !((st_flags & LS_SETTLED) && may_park);
    }
    if (!wave_any(want)) return;
  }
  LaneLds LL;
  {
    const int nopt = P.max_allele > 1 ? P.max_allele - 1 : 1;
    LL.win_n = lane_window(mmax);
    LL.tri = spec_memo_entries(mmax);
    LL.ndraws = spec_draws(KT, mmax);
    unsigned char *p = smem;
    LL.win = lds_cast<uint64_t>(p); p += (size_t)8 * NC * LL.win_n;
    LL.bcum = lds_cast<uint64_t>(p); p += (size_t)8 * NC * mmax;
    LL.tbuf = lds_cast<uint64_t>(p); p += (size_t)8 * NC * LANE_TB * (KT + 1);
    LL.gptr = lds_cast<uint64_t>(p); p += (size_t)8 * NC * GP_N;
    LL.gval = lds_cast<double>(p); p += (size_t)8 * NC * GV_N;
    LL.prior = lds_cast<double>(p); p += (size_t)8 * NC * (2 * KT + 5);
    LL.gstream = lds_cast<uint32_t>(p); p += (size_t)16 * NC;
    LL.memo = lds_cast<uint32_t>(p); p += (size_t)4 * NC * 2 * LL.tri;
    LL.cols = lds_cast<uint16_t>(p); p += (size_t)2 * NC * mmax;
    LL.shift = lds_cast<uint8_t>(p); p += (size_t)NC * mmax;
    LL.nal = lds_cast<uint8_t>(p); p += (size_t)NC * mmax;
    LL.ord = lds_cast<uint8_t>(p); p += (size_t)NC * (mmax + 1);
    p = smem + (((size_t)(p - smem) + 1) & ~(size_t)1);
    LL.nreads = lds_cast<uint16_t>(p); p += (size_t)2 * NC;
    LL.ndict = lds_cast<uint16_t>(p); p += (size_t)2 * NC;
    p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
    LL.pw = lds_cast<uint64_t>(p); p += (size_t)8 * KT * 64;
    LL.ptab = lds_cast<double>(p); p += (size_t)8 * 64;
    LL.optp = lds_cast<double>(p); p += (size_t)8 * nopt * 64;
    LL.optl = lds_cast<double>(p); p += (size_t)8 * nopt * 64;
    LL.ln = lds_cast<double>(p); p += (size_t)8 * SPEC_LN;
    LL.lninv = lds_cast<double>(p); p += (size_t)8 * SPEC_LN;
    LL.draws = lds_cast<uint64_t>(p); p += (size_t)8 * LL.ndraws;
    LL.tab_bytes = P.max_ma * WAVE * P.cstride;
    LL.tcw = lds_cast<double>(p); p += (size_t)8 * NC * rpad;
    LL.dict = lds_cast<double>(p); p += (size_t)8 * NC * DICT_MAX;
    LL.tct = lds_cast<uint8_t>(p); p += (size_t)NC * LL.tab_bytes;
    LL.bw = lds_cast<uint64_t>(p); p += (size_t)8 * KT;
    LL.cum = lds_cast<int>(p); p += (size_t)4 * 64;
    LL.permtab = lds_cast<uint16_t>(p); p += (size_t)2 * nmax;
    LL.itab = lds_cast<uint16_t>(p); p += (size_t)2 * LL.tri;
    LL.ktab = lds_cast<uint8_t>(p); p += (size_t)nmax;
  }
  for (int i = lane; i < SPEC_LN; i += WAVE) {
    LL.ln[i] = c_ln[i];
    LL.lninv[i] = c_ln_inv[i];
  }
  for (int i = lane; i < LL.tri; i += WAVE) {  // inverse of spec_memo_index
    int stop = 1;
    while (stop * (stop + 1) / 2 <= i) stop++;
    const int start = i - stop * (stop - 1) / 2;
    LL.itab[i] = (uint16_t)(start | (stop << 8));
  }
  LChain<KT> c;
  c.alive = q < n_chains;
  const int u = c.alive ? (int)(q / Cn) : 0;
  const int chain = c.alive ? (int)(q % Cn) : 0;
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);
  if (c.alive && mi[META_I_STATUS] != MCHAP_UNIT_OK) c.alive = false;
  const int A = U.max_allele;
  c.Mh = c.alive ? mi[META_I_MH] : 1;
  c.bits = allele_bits(A);
  const int Mh = c.Mh;
  const int tri_h = spec_memo_entries(Mh);
  const bool cache_on = D.cache_slots > 0;
  uint32_t *gmemo = reinterpret_cast<uint32_t *>(P.lane_memo) + (size_t)(c.alive ? q : 0) * 2 * LL.tri;
  if (sl == 0) {
    LL.gval[ci * GV_N + GV_INB] = U.inbreeding;
    LL.gval[ci * GV_N + GV_MLO] = 0.0;
    LL.gval[ci * GV_N + GV_MHI] = 0.0;
    LL.gstream[ci * 4 + 0] = (uint32_t)D.seed;
    LL.gstream[ci * 4 + 1] = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
    LL.gstream[ci * 4 + 2] = ((uint32_t)chain << 16) | 0u;
    LL.gstream[ci * 4 + 3] = (uint32_t)U.stream_id;
    LDSP(uint64_t) gp = LL.gptr + ci * GP_N;
    gp[GP_RT] = (uint64_t)(uintptr_t)(P.rt + (size_t)u * P.max_ma * rpad);
    gp[GP_CW] = (uint64_t)(uintptr_t)(P.cntw + (size_t)u * rpad);
    gp[GP_CT] = (uint64_t)(uintptr_t)(P.codes + (size_t)u * P.max_ma * WAVE * P.cstride);
    gp[GP_CACHE] = cache_on ? (uint64_t)(uintptr_t)(reinterpret_cast<ulonglong2 *>(D.cache) + (size_t)q * (size_t)D.cache_slots) : 0ull;
    gp[GP_CKEYS] = (cache_on && D.cache_keys) ? (uint64_t)(uintptr_t)(D.cache_keys + (size_t)q * (size_t)D.cache_slots * D.cache_key_words) : 0ull;
    gp[GP_TRACE] = (uint64_t)(uintptr_t)(D.trace + U.trace_off + (size_t)chain * D.steps * KT);
    gp[GP_LLK] = (uint64_t)(uintptr_t)(D.llks + U.llk_off + (size_t)chain * D.steps);
    LL.nreads[ci] = (uint16_t)(c.alive ? U.n_reads : 0);
    LL.ndict[ci] = (uint16_t)(((c.alive && !(P.flags & 4)) ? mi[META_I_NDICT] : 0) | ((c.alive && mi[META_I_W01] != 0) ? (int)ND_W01 : 0));
  }
  if (c.alive) {
    for (int j = sl; j < Mh; j += L) {
      LL.cols[(size_t)ci * mmax + j] = (uint16_t)mi[META_I_COLS + j];
      LL.nal[(size_t)ci * mmax + j] = (uint8_t)mi[META_I_COLS + P.max_pos + j];
      LL.shift[(size_t)ci * mmax + j] = (uint8_t)(c.bits * (Mh - 1 - j));
    }
    if (!isnan(U.inbreeding))
      for (int i = sl; i < 2 * KT + 5; i += L) LL.prior[(size_t)ci * (2 * KT + 5) + i] = mf[meta_f_prior(0) + i];
    if (D.n_intervals == 0 && sl == 0) {
      // cumulative break-count distribution, summed in the reference's order (structural.py:44-49), as thresholds
      double cacc = 0.0;
      for (int j = 0; j < Mh; j++) {
        cacc += D.break_table[(size_t)Mh * D.max_pos + j];
        LL.bcum[ci * mmax + j] = ceil53(cacc);
      }
    }
    if (resume)
      for (int i = sl; i < 2 * LL.tri; i += L) LL.memo[(size_t)ci * 2 * LL.tri + i] = gmemo[i];
    else
      for (int i = sl; i < 2 * LL.tri; i += L) LL.memo[(size_t)ci * 2 * LL.tri + i] = MEMO_UNKNOWN;
  }
  // the unit tables of the wave's chains: coded table, read weights, dictionary (a chain that will not run here -- a
  // settled one on resume -- still gets its slot filled: cheap, and keeps the code uniform)
  for (int cs = 0; cs < NC; cs++) {
    const int owner = cs << lsh;
    const bool live = __builtin_amdgcn_readlane((int)c.alive, owner) != 0;
    if (!live) continue;
    const int unit = __builtin_amdgcn_readlane(u, owner);
    const int nd = __builtin_amdgcn_readfirstlane((int)((P.flags & 4) ? 0 : P.meta_i[(size_t)unit * meta_i_stride(P.max_pos) + META_I_NDICT]));
    const uint32_t *src = reinterpret_cast<const uint32_t *>(P.codes + (size_t)unit * LL.tab_bytes);
    LDSP(uint32_t) dst = (LDSP(uint32_t))(LL.tct + (size_t)cs * LL.tab_bytes);
    if (nd > 0)
      for (int i = lane; i < LL.tab_bytes / 4; i += WAVE) dst[i] = src[i];
    const double *cws = P.cntw + (size_t)unit * rpad;
    for (int i = lane; i < rpad; i += WAVE) LL.tcw[(size_t)cs * rpad + i] = cws[i];
    const double *du = P.dict + (size_t)unit * DICT_MAX;
    for (int i = lane; i < nd; i += WAVE) LL.dict[(size_t)cs * DICT_MAX + i] = du[i];
  }
  lds_sync();
  const int amax = [&] {
    int v = c.alive ? A : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
    return v;
  }();
  {
    GWords<KT> z;
#pragma unroll
    for (int h = 0; h < KT; h++) z.w[h] = 0;
    c.g = z;
  }
  c.ctr = 0;
  c.doff = 0;
  c.dcount = 0;
  c.llk = 0.0;
  c.lo = 0;
  c.hi = 0;
  c.mvalid = false;
  c.stable = false;
  c.n_unknown = c.alive ? 2 * tri_h : 0;
  c.cursor = 0;
  int my_step = 0;   // the chain's next step
  int skip = 0;      // first compound step to run in the chain's first step here (resume in mid-step)
  bool running = c.alive;
  if (resume) {
    if (c.alive) {
      const LaneState st = state[q];
      my_step = st.step;
      skip = st.phase;
      running = st.step < Sn && !((st.flags & LS_SETTLED) && may_park);
#pragma unroll
      for (int h = 0; h < KT; h++) set_word<KT>(c.g, h, st.g[h]);
      c.llk = st.llk;
      c.ctr = st.ctr;
      c.lo = st.lo;
      c.hi = st.hi;
      c.mvalid = (st.flags & LS_MVALID) != 0;
      c.stable = true;
    }
  } else if (c.alive) {
    // ---- initial genotype (assemble/mcmc.py:202-208) ----
    LDSP(uint8_t) shift = LL.shift + ci * mmax;
    if (U.initial_off >= 0) {
      const int8_t *ini = D.initial + U.initial_off + (size_t)chain * KT * Mh;
#pragma unroll
      for (int h = 0; h < KT; h++) {
        uint64_t x = 0;
        for (int j = 0; j < Mh; j++) x |= (uint64_t)(uint8_t)ini[h * Mh + j] << shift[j];
        set_word<KT>(c.g, h, x);
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
  const bool memo_mut = !(P.flags & 1), memo_iv = !(P.flags & 2);
  int status = MCHAP_UNIT_OK;
  // exact count of the chain's unknown thresholds (wave-level: every lane must call)
  auto recount = [&]() {
    int n = 0;
    if (c.alive)
      for (int i = sl; i < 2 * tri_h; i += L) {
        const int ty = i >= tri_h ? 1 : 0;
        n += LL.memo[(size_t)ci * 2 * LL.tri + (size_t)ty * LL.tri + (i - ty * tri_h)] == MEMO_UNKNOWN ? 1 : 0;
      }
    for (int o = 1; o < L; o <<= 1) n += __shfl_xor(n, o, WAVE);
    return n;
  };
  if (resume) c.n_unknown = recount();

  // Wave-uniform view of chain `cs` for the serving code (speculative-kernel context, group 0 == the chain).
  auto open_chain = [&](int cs, Grp<KT> &cu, SpecLds &S) {
    const int owner = cs << lsh;
    lane_context<KT>(LL, cs, mmax, S);
    S.cache_on = cache_on;
    S.cache_mask = cache_on ? (uint32_t)(D.cache_slots / 8) - 1u : 0u;
    S.key_words = D.cache_key_words;
    S.reuse_on = !(P.flags & 8);
    S.crow = WAVE * P.cstride;
    cu.Mh = __builtin_amdgcn_readlane(c.Mh, owner);
    cu.bits = __builtin_amdgcn_readlane(c.bits, owner);
    cu.alive = true;
    cu.flat = false;  // (the shortcut of denovo_spec_kernel for all-gap units is not used by this kernel)
    cu.ctr = bcast_u64(c.ctr, owner);
    cu.doff = 0;
    cu.dcount = 0;
    cu.llk = bcast_f64(c.llk, owner);
    cu.g = bcast_words<KT>(c.g, owner);
    cu.memo_on = memo_mut;
    cu.mvalid = false;
    cu.mwin = 8;
    cu.gen = 1;
    cu.memo_gen = 1;
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
    for (int i_ = 0; i_ < 12; i_++) cu.ph[i_] = 0;
    cu.pt0 = __builtin_amdgcn_s_memtime();
#endif
    lane_context_tabs<KT>(LL, cs, rpad, S);
  };
  auto wipe_memo = [&](int cs) {
    for (int i = lane; i < 2 * LL.tri; i += WAVE) LL.memo[(size_t)cs * 2 * LL.tri + i] = MEMO_UNKNOWN;
  };

  // ---- initial likelihood (assemble/mcmc.py:303): one evaluation per chain, the wave serving chain after chain ----
  if (!resume) {
    unsigned long long todo = __ballot(c.alive && sl == 0);
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int cs = owner >> lsh;
      Grp<KT> cu;
      SpecLds S;
      open_chain(cs, cu, S);
      const GWords<KT> g0 = cu.g;
      if (lane == 0) {
#pragma unroll
        for (int h = 0; h < KT; h++) S.pw[h * WAVE] = g0.w[h];
      }
      lds_sync();
      const double v = spec_coop_all<KT, 64, true>(1ull, S.pw, S.shift, S.cols, S.nreads, S.ndict, S.dict, S.gptr, S.bw, false, S.crow, mmax, cu.Mh,
                                                   (1u << cu.bits) - 1u, rpad, lane, S.lds_ct, S.lds_cw);
      lds_sync();
      const double l0 = __shfl(v, 0, WAVE);
      if (ci == cs) c.llk = l0;
    }
  }

#if defined(MCHAP_PHASES)
  unsigned long long lph[24];
  for (int i_ = 0; i_ < 24; i_++) lph[i_] = 0;
  unsigned long long lpt0 = __builtin_amdgcn_s_memtime();
#endif
  int nbuf = 0;  // trace records of the chain waiting in LDS: steps my_step - nbuf .. my_step - 1
  const uint64_t pthr0 = decision_threshold(D.p_recomb), pthr1 = decision_threshold(D.p_partial), pthr2 = decision_threshold(D.p_dosage);
  while (wave_any(running)) {
    LPH(0);
    if (running && isnan(c.llk)) {  // assemble/mcmc.py:330-331
      status = MCHAP_UNIT_NAN_LLK;
      c.alive = false;
      running = false;
    }
    bool changed = false;  // genotype changed during this step
    // =============================== mutation compound step ===============================
    {
      const bool act = running && skip == 0;
      const int n = KT * Mh;
      if (act) c.ctr = (uint64_t)my_step * STEP_DRAWS;  // the step's draws (philox.hpp)
      const uint64_t ctr0 = c.ctr;
      const int E = lane_extra(Mh);
      LDSP(uint64_t) win = LL.win + (size_t)ci * LL.win_n;
      bool ok = act && c.mvalid;
      if (ok) ok = fast_mutation(ld_stream_l(LL.gstream, ci), ctr0 + (uint64_t)(n - 1), n, E, c.lo, c.hi, win, sl, L);
      lds_sync();  // the window entries were written by the chain's other lanes
      {
        const unsigned long long badm = __ballot(act && !ok);
        const unsigned long long mine = L == 64 ? ~0ull : (((1ull << L) - 1ull) << (ci * L));
        ok = act && !(badm & mine);
      }
      if (ok) {
        c.ctr = ctr0 + (uint64_t)(2 * n - 1);
        c.doff = 0;
        c.dcount = E;
      }
      LPH(1);
      // chains that could not decide: the whole wave runs the step for them, one chain at a time
      unsigned long long todo = __ballot(act && !ok && sl == 0);
      LCNT(10, __popcll(todo));
      while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cs = owner >> lsh;
        Grp<KT> cu;
        SpecLds S;
        open_chain(cs, cu, S);
        const GWords<KT> before = cu.g;
        spec_mutation<KT, 64, true>(cu, S, 1.0, amax, mmax, nmax, rpad, lane, 0, lane);
        LCNT(20, cu.ph[9]);
        LCNT(21, cu.ph[10]);
        LCNT(22, cu.ph[11]);
        bool diff = false;
#pragma unroll
        for (int h = 0; h < KT; h++) diff = diff || before.w[h] != cu.g.w[h];
        lds_sync();
        const double mlo = S.gval[GV_MLO], mhi = S.gval[GV_MHI];
        if (ci == cs) {
          c.g = cu.g;
          c.llk = cu.llk;
          c.ctr = cu.ctr;
          c.doff = 0;
          c.dcount = 0;  // nothing staged behind the step
          c.mvalid = cu.mvalid;
          c.lo = ceil53(mlo);
          c.hi = ceil53(mhi);
          if (diff) {
            changed = true;
            c.n_unknown = 2 * tri_h;
          }
        }
        if (diff) wipe_memo(cs);  // the interval-step thresholds described the previous genotype
        lds_sync();
      }
    }
    LPH(2);
    // =============================== structural compound steps ===============================
#pragma unroll 1
    for (int kind = 0; kind < 3; kind++) {
      const int step_type = kind == 0 ? 0 : 1;
      LDSP(uint64_t) win = LL.win + (size_t)ci * LL.win_n;
      LDSP(uint32_t) memo = LL.memo + (size_t)ci * 2 * LL.tri + (size_t)step_type * LL.tri;
      const uint64_t pthr = kind == 0 ? pthr0 : (kind == 1 ? pthr1 : pthr2);
      bool pending = running && skip <= kind + 1;  // chain has not finished this compound step
      bool exact = false;                          // thresholds could not decide: visiting order + walk
      uint64_t zeros = 0;
      int n_int = 0;
      int doff1 = 0;  // window offset after decision / breaks (start of the permutation draws)
      // ---- fast path; re-run after a refill for the chains whose window ran out ----
      while (wave_any(pending && !exact)) {
        LCNT(11, 1);
        const bool act = pending && !exact;
        SFast F;
        F.short_ = false;
        if (act) F = fast_structural_ll(kind, pthr, D.n_intervals, Mh, win, c.doff, c.dcount, LL.bcum + (size_t)ci * mmax, memo, memo_iv);
        if (act && !F.short_) {
          if (F.bad) {
            status = MCHAP_UNIT_BREAKS;
            c.alive = false;
            running = false;
            pending = false;
          } else if (F.done) {
            c.ctr += (uint64_t)(F.doff_end - c.doff);
            c.doff = F.doff_end;
            pending = false;
          } else {
            exact = true;  // c.ctr / c.doff stay at the start of the step; doff1 marks the permutation draws
            zeros = F.zeros;
            n_int = F.n_int;
            doff1 = F.doff1;
          }
        }
        // refill the windows that ran out (from the chain's current draw) and go again
        if (wave_any(act && F.short_)) {
          LCNT(12, 1);
          const bool mine = act && F.short_;
          if (mine) stage_draws_l(ld_stream_l(LL.gstream, ci), c.ctr, LL.win_n, win, sl, L);
          lds_sync();
          if (mine) {
            c.doff = 0;
            c.dcount = LL.win_n;
          }
        }
      }
      LPH(3);
      // ---- exact path: visiting order (np.random.permutation), then the intervals one after the other ----
      if (wave_any(exact)) {
        LCNT(13, 1);
        LDSP(uint8_t) ord = LL.ord + (size_t)ci * (mmax + 1);
        // the window must hold the rest of the step: n_int - 1 shuffle draws + up to n_int uniforms
        if (wave_any(exact && c.dcount - doff1 < 2 * n_int)) {
          const bool mine = exact && c.dcount - doff1 < 2 * n_int;
          if (mine) stage_draws_l(ld_stream_l(LL.gstream, ci), c.ctr, LL.win_n, win, sl, L);
          lds_sync();
          if (mine) {
            doff1 -= c.doff;
            c.doff = 0;
            c.dcount = LL.win_n;
          }
        }
        int doff = doff1;
        if (exact) {
          for (int i = sl; i < n_int; i += L) ord[i] = (uint8_t)i;
        }
        lds_sync();
        if (exact) {
          for (int i = n_int - 1; i >= 1; i--) {
            const int k = (int)__umulhi((uint32_t)win[doff], (uint32_t)(i + 1));
            doff++;
            const uint8_t a = ord[i], b = ord[k];
            ord[i] = b;
            ord[k] = a;
          }
        }
        lds_sync();
        int qi = 0;
        while (wave_any(exact)) {
          // advance over the intervals the thresholds decide
          int req_idx = 0;
          if (exact) {
            while (qi < n_int) {
              const int iv = ord[qi];
              uint64_t z = zeros;
              for (int r = 0; r < iv; r++) z &= z - 1;
              const int start = __ffsll((long long)z) - 1;
              z &= z - 1;
              const int stop = __ffsll((long long)z) - 1;
              req_idx = spec_memo_index(start, stop);
              const uint32_t t = memo_iv ? memo[req_idx] : MEMO_UNKNOWN;
              if (t == MEMO_NOOPT) {
                qi++;
                continue;
              }
              if (t != MEMO_UNKNOWN && (((uint32_t)win[doff]) >> 5) >= t) {
                doff++;
                qi++;
                continue;
              }
              break;
            }
            if (qi >= n_int) {
              c.ctr += (uint64_t)(doff - c.doff);
              c.doff = doff;
              exact = false;
              pending = false;
            }
          }
          // serve the blocked chains, one at a time
          unsigned long long todo = __ballot(exact && sl == 0);
          LCNT(14, __popcll(todo));
          while (todo) {
            const int owner = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int cs = owner >> lsh;
            Grp<KT> cu;
            SpecLds S;
            open_chain(cs, cu, S);
            const int r_idx = __builtin_amdgcn_readlane(req_idx, owner);
            const int r_doff = __builtin_amdgcn_readlane(doff, owner);
            const uint64_t uw = LL.win[(size_t)cs * LL.win_n + r_doff];
            const int cur = __builtin_amdgcn_readlane(c.cursor, owner);
            const ServeResult<KT> R = serve_structural<KT>(cu, S, LL, cs, true, step_type, r_idx, uw, cur, mmax, rpad, lane);
            LCNT(8, cu.ph[9]);
            LCNT(9, cu.ph[10]);
            LCNT(23, cu.ph[11]);
            if (ci == cs) {
              if (R.n_opt_req > 0) doff++;  // the step's uniform (structural.py:504-506: none without options)
              if (R.moved) {
                c.g = cu.g;
                c.llk = cu.llk;
                c.mvalid = false;
                changed = true;
                c.n_unknown = 2 * tri_h;
              } else {
                c.n_unknown -= R.filled;
                c.cursor += WAVE - 1;
                if (c.cursor >= 2 * tri_h) c.cursor %= 2 * tri_h;
              }
              qi++;
            }
            if (R.moved) {  // the thresholds described the previous genotype
              wipe_memo(cs);
              lds_sync();
            }
          }
        }
      }
      LPH(7);
    }
    LPH(4);
    // =============================== thresholds of settled chains ===============================
    // a chain whose genotype survived the previous and this step gets all its unknown thresholds now
    if (memo_iv && wave_any(running && c.n_unknown > 0 && !changed && c.stable)) {
      unsigned long long todo = __ballot(running && c.n_unknown > 0 && !changed && c.stable && sl == 0);
      LCNT(15, __popcll(todo));
      while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cs = owner >> lsh;
        Grp<KT> cu;
        SpecLds S;
        open_chain(cs, cu, S);
        int cur = __builtin_amdgcn_readlane(c.cursor, owner);
        const int total = 2 * spec_memo_entries(cu.Mh);
        int left = total;  // upper bound of the rounds: every round resolves at least one candidate it finds
        for (int pass = 0; pass < total && left > 0; pass++) {
          const ServeResult<KT> R = serve_structural<KT>(cu, S, LL, cs, false, 0, 0, 0ull, cur, mmax, rpad, lane);
          cur += WAVE;
          if (cur >= total) cur %= total;
          // done when a whole sweep over the table found nothing (counted exactly below)
          int n = 0;
          for (int i = lane; i < total; i += WAVE) {
            const int ty = i >= total / 2 ? 1 : 0;
            n += LL.memo[(size_t)cs * 2 * LL.tri + (size_t)ty * LL.tri + (i - ty * (total / 2))] == MEMO_UNKNOWN ? 1 : 0;
          }
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, WAVE);
          left = n;
          (void)R;
        }
        if (ci == cs) {
          c.n_unknown = left;
          c.cursor = cur;
        }
      }
    }
    LPH(5);
    // =============================== record ===============================
    {
      LDSP(uint64_t) tb = LL.tbuf + (size_t)ci * LANE_TB * (KT + 1);
      if (running) {
        const GWords<KT> gr = c.g;
        for (int w = sl; w < KT; w += L) {
          const uint64_t x = sel_word<KT>(gr, w);
          int rank = 0;
#pragma unroll
          for (int h = 0; h < KT; h++) rank += (gr.w[h] < x || (gr.w[h] == x && h < w)) ? 1 : 0;
          tb[nbuf * KT + rank] = x;
        }
        if (sl == 0) tb[LANE_TB * KT + nbuf] = (uint64_t)__double_as_longlong(c.llk);
        nbuf++;
        my_step++;
        skip = 0;
      }
      c.stable = !changed;
      // finished, or settled: every threshold known, bounds valid, genotype stable -> the steady kernel takes over
      bool stop = false, park = false;
      if (running && my_step >= Sn) stop = true;
      if (running && !stop && may_park && memo_iv && memo_mut && c.mvalid && c.n_unknown == 0 && c.stable) {
        stop = true;
        park = true;
      }
      const bool flush = running && (nbuf == LANE_TB || stop);
      if (wave_any(flush)) {
        lds_sync();
        if (flush) {
          const int first = my_step - nbuf;
          const int nw = nbuf * KT;
          uint64_t *tp = reinterpret_cast<uint64_t *>((uintptr_t)LL.gptr[ci * GP_N + GP_TRACE]) + (size_t)first * KT;
          for (int i = sl; i < nw; i += L) tp[i] = tb[i];
          uint64_t *lp = reinterpret_cast<uint64_t *>((uintptr_t)LL.gptr[ci * GP_N + GP_LLK]) + first;
          for (int i = sl; i < nbuf; i += L) lp[i] = tb[LANE_TB * KT + i];
          nbuf = 0;
        }
        lds_sync();
      }
      LCNT(16, __popcll(__ballot(park && sl == 0)));
      LCNT(17, wave_sum_i((park && sl == 0) ? my_step : 0));
      LCNT(19, __popcll(__ballot(stop && !park && sl == 0)));
      if (stop) {
        running = false;
        if (park) {
          // hand the chain over: state record + its threshold table
          for (int i = sl; i < 2 * LL.tri; i += L) gmemo[i] = LL.memo[(size_t)ci * 2 * LL.tri + i];
          if (sl == 0) {
            LaneState st;
            const GWords<KT> gr = c.g;
#pragma unroll
            for (int h = 0; h < MCHAP_MAX_PLOIDY; h++) st.g[h] = h < KT ? gr.w[h < KT ? h : 0] : 0ull;
            st.llk = c.llk;
            st.ctr = c.ctr;
            st.lo = c.lo;
            st.hi = c.hi;
            st.step = my_step;
            st.phase = 0;
            st.flags = LS_MVALID | LS_SETTLED;
            st.pad = 0;
            state[q] = st;
          }
        } else if (sl == 0) {
          state[q].step = Sn;
        }
      }
    }
  }
  // dead chains (errors) and lanes beyond the batch: nothing left to run
  if (q < n_chains && !c.alive && sl == 0) state[q].step = Sn;
#if defined(MCHAP_PHASES)
  LPH(6);
  if (threadIdx.x == 0)
    for (int i_ = 0; i_ < 24; i_++) atomicAdd(&g_stats[i_], lph[i_]);
#endif
  if (status != MCHAP_UNIT_OK && sl == 0) atomicMax(&D.status[u], status);
}

// ---------------------------------------------------------------------------------------------------------
// Steady kernel: settled chains on thresholds alone.  No likelihood code, few registers, little LDS: many waves per
// SIMD.  A chain runs until a compound step cannot be decided (it is handed back at the start of that compound step)
// or to the end; its genotype cannot change here, so its trace rows are written in one sweep when it stops.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t steady_lds_bytes(int Mmax, int L) {
  const int NC = 64 / L;
  size_t b = 0;
  b += (size_t)8 * NC * lane_window(Mmax);
  b += (size_t)8 * NC * Mmax;
  b += (size_t)4 * NC * 2 * spec_memo_entries(Mmax);
  return (b + 63) & ~(size_t)63;
}

template <int KT>
__global__ __launch_bounds__(64, 4) void denovo_steady_kernel(const SimtParams P, const int lsh) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int L = 1 << lsh;
  const int NC = WAVE >> lsh;
  const int lane = threadIdx.x;
  const int ci = lane >> lsh, sl = lane & (L - 1);
  const int Cn = D.chains, Sn = D.steps;
  const int mmax = P.max_pos;
  const int win_n = lane_window(mmax), tri = spec_memo_entries(mmax);
  const long long q = (long long)blockIdx.x * NC + ci;
  const long long n_chains = (long long)P.n_units * Cn;
  LaneState *state = reinterpret_cast<LaneState *>(P.lane_state);
  bool running = false;
  int my_step = 0;
  if (q < n_chains) {
    my_step = state[q].step;
    running = my_step < Sn && (state[q].flags & LS_SETTLED);
  }
  if (!wave_any(running)) return;
  const bool ran = running;
  LDSP(uint64_t) win = lds_cast<uint64_t>(smem) + (size_t)ci * win_n;
  LDSP(uint64_t) bc = lds_cast<uint64_t>(smem + (size_t)8 * NC * win_n) + (size_t)ci * mmax;
  LDSP(uint32_t) memo = lds_cast<uint32_t>(smem + (size_t)8 * NC * win_n + (size_t)8 * NC * mmax) + (size_t)ci * 2 * tri;
  const int u = running ? (int)(q / Cn) : 0;
  const int chain = running ? (int)(q % Cn) : 0;
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const int Mh = running ? mi[META_I_MH] : 1;
  uint64_t ctr = 0, lo = 0, hi = 0;
  Stream st;
  st.k0 = (uint32_t)D.seed;
  st.k1 = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
  st.c2 = ((uint32_t)chain << 16) | 0u;
  st.c3 = (uint32_t)U.stream_id;
  if (running) {
    ctr = state[q].ctr;
    lo = state[q].lo;
    hi = state[q].hi;
    const uint32_t *gm = reinterpret_cast<const uint32_t *>(P.lane_memo) + (size_t)q * 2 * tri;
    for (int i = sl; i < 2 * tri; i += L) memo[i] = gm[i];
    if (D.n_intervals == 0 && sl == 0) {
      double cacc = 0.0;
      for (int j = 0; j < Mh; j++) {
        cacc += D.break_table[(size_t)Mh * D.max_pos + j];
        bc[j] = ceil53(cacc);
      }
    }
  }
  lds_sync();
  const int s_begin = my_step;
  int phase = 0;  // compound step the chain stopped at (when it stops before the end)
  int doff = 0, dcount = 0;
  const int n = KT * Mh, E = lane_extra(Mh);
  const uint64_t pthr0 = decision_threshold(D.p_recomb), pthr1 = decision_threshold(D.p_partial), pthr2 = decision_threshold(D.p_dosage);
  const unsigned long long mine_mask = L == 64 ? ~0ull : (((1ull << L) - 1ull) << (ci * L));
  while (wave_any(running)) {
    // ---- mutation compound step ----
    bool ok = running;
    if (running) ctr = (uint64_t)my_step * STEP_DRAWS;  // the step's draws (philox.hpp)
    if (ok && !(P.flags & 16)) ok = fast_mutation(st, ctr + (uint64_t)(n - 1), n, E, lo, hi, win, sl, L);  // (flag: timing experiments)
    lds_sync();
    ok = running && !(__ballot(running && !ok) & mine_mask);
    if (running && !ok) {
      running = false;
      phase = 0;
    }
    if (running) {
      ctr += (uint64_t)(2 * n - 1);
      doff = 0;
      dcount = E;
    }
    // ---- structural compound steps ----
#pragma unroll 1
    for (int kind = 0; kind < 3; kind++) {
      const uint64_t pthr = kind == 0 ? pthr0 : (kind == 1 ? pthr1 : pthr2);
      LDSP(uint32_t) mt = memo + (size_t)(kind == 0 ? 0 : 1) * tri;
      bool pending = running && !(P.flags & 32);
      while (wave_any(pending)) {
        SFast F;
        F.short_ = false;
        if (pending) F = fast_structural_ll(kind, pthr, D.n_intervals, Mh, win, doff, dcount, bc, mt, true);
        if (pending && !F.short_) {
          if (F.done) {
            ctr += (uint64_t)(F.doff_end - doff);
            doff = F.doff_end;
          } else {  // cannot be decided here (or the reference's "breaks" error): back to the settling kernel
            running = false;
            phase = kind + 1;
          }
          pending = false;
        }
        if (wave_any(pending && F.short_)) {
          if (pending) stage_draws_l(st, ctr, win_n, win, sl, L);
          lds_sync();
          if (pending) {
            doff = 0;
            dcount = win_n;
          }
        }
      }
    }
    if (running) {
      my_step++;
      if (my_step >= Sn) running = false;
    }
  }
  // ---- the chain's rows of the trace: one record, my_step - s_begin times ----
  if (ran && my_step > s_begin) {
    const LaneState &stt = state[q];
    uint64_t gw[KT];
#pragma unroll
    for (int h = 0; h < KT; h++) gw[h] = stt.g[h];
    // canonical (ascending) order
#pragma unroll
    for (int a = 1; a < KT; a++) {
#pragma unroll
      for (int b = 0; b < KT - 1; b++) {
        if (b < KT - a) {
          const uint64_t x = gw[b], y = gw[b + 1];
          gw[b] = x < y ? x : y;
          gw[b + 1] = x < y ? y : x;
        }
      }
    }
    const unsigned long long llk_bits = (unsigned long long)__double_as_longlong(stt.llk);
    uint64_t *tp = D.trace + U.trace_off + ((size_t)chain * D.steps + s_begin) * KT;
    const int nw = (my_step - s_begin) * KT;
    for (int i = sl; i < nw; i += L) {
      const int h = i % KT;
      uint64_t x = gw[0];
#pragma unroll
      for (int k = 1; k < KT; k++) x = (h == k) ? gw[k] : x;
      tp[i] = x;
    }
    uint64_t *lp = reinterpret_cast<uint64_t *>(D.llks + U.llk_off + (size_t)chain * D.steps + s_begin);
    for (int i = sl; i < my_step - s_begin; i += L) lp[i] = llk_bits;
  }
#if defined(MCHAP_PHASES)
  if (ran && sl == 0) {
    atomicAdd(&g_stats[18], (unsigned long long)(my_step - s_begin));
    atomicAdd(&g_stats[20], 1ull);
    if (my_step < Sn) atomicAdd(&g_stats[21], 1ull);
  }
#endif
  if (ran && sl == 0) {
    LaneState &stt = state[q];
    stt.ctr = ctr;
    stt.step = my_step;
    stt.phase = phase;
    if (my_step < Sn) stt.flags &= ~LS_SETTLED;  // stopped before the end: the settling kernel resumes it
  }
}

}  // namespace mchap
