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
  LDSP(double) dict;       // [DICT_MAX]
  LDSP(uint64_t) bw;       // [K]
  LDSP(int) cum;           // [64]
  LDSP(uint16_t) itab;     // [tri] start | stop << 8 of triangular index
  int win_n, tri, ndraws;
};

__host__ __device__ inline size_t lane_lds_bytes(int K, int Mmax, int Amax, int L) {
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
  b += (size_t)8 * DICT_MAX;
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
  S.dict = LL.dict;
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

// The unit's dictionary into the wave's LDS slot (the co-operative evaluation gathers its factors from it).
__device__ __forceinline__ void load_dict(const LaneLds &LL, const double *du, int nd, int lane) {
  for (int i = lane; i < nd; i += WAVE) LL.dict[i] = du[i];
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
  const double llk_i = spec_eval<KT, 64>(prop, pw, cu, S, mmax, rpad, lane);
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

template <int KT>
__global__ __launch_bounds__(64, 2) void denovo_lane_kernel(const SimtParams P, const int lsh) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int L = 1 << lsh;
  const int NC = WAVE >> lsh;
  const int lane = threadIdx.x;
  const int ci = lane >> lsh, sl = lane & (L - 1);
  const int Cn = D.chains, Sn = D.steps;
  const int mmax = P.max_pos, nmax = KT * P.max_pos;
  const int rpad = D.rpad;
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
    LL.dict = lds_cast<double>(p); p += (size_t)8 * DICT_MAX;
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
  const long long q = (long long)blockIdx.x * NC + ci;  // chain index
  const long long n_chains = (long long)P.n_units * Cn;
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
  const bool cache_on = D.cache_slots > 0;
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
    gp[GP_TRACE] = (uint64_t)(uintptr_t)(D.trace + U.trace_off + (size_t)chain * D.steps * KT);
    gp[GP_LLK] = (uint64_t)(uintptr_t)(D.llks + U.llk_off + (size_t)chain * D.steps);
    LL.nreads[ci] = (uint16_t)(c.alive ? U.n_reads : 0);
    LL.ndict[ci] = (uint16_t)((c.alive && !(P.flags & 4)) ? mi[META_I_NDICT] : 0);
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
    for (int i = sl; i < 2 * LL.tri; i += L) LL.memo[(size_t)ci * 2 * LL.tri + i] = MEMO_UNKNOWN;
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
  if (c.alive) {
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
  c.ctr = 0;
  c.doff = 0;
  c.dcount = 0;
  c.llk = 0.0;
  c.lo = 0;
  c.hi = 0;
  c.mvalid = false;
  c.stable = false;
  c.n_unknown = c.alive ? 2 * spec_memo_entries(Mh) : 0;
  c.cursor = 0;
  const bool memo_mut = !(P.flags & 1), memo_iv = !(P.flags & 2);
  int cur_dict_unit = -1;  // unit whose dictionary sits in LL.dict (wave-uniform)
  int status = MCHAP_UNIT_OK;

  // Wave-uniform view of chain `cs` for the serving code (speculative-kernel context, group 0 == the chain).
  auto open_chain = [&](int cs, Grp<KT> &cu, SpecLds &S) {
    const int owner = cs << lsh;
    lane_context<KT>(LL, cs, mmax, S);
    S.cache_on = cache_on;
    S.cache_mask = cache_on ? (uint32_t)(D.cache_slots / 8) - 1u : 0u;
    S.reuse_on = !(P.flags & 8);
    S.crow = WAVE * P.cstride;
    cu.Mh = __builtin_amdgcn_readlane(c.Mh, owner);
    cu.bits = __builtin_amdgcn_readlane(c.bits, owner);
    cu.alive = true;
    cu.ctr = bcast_u64(c.ctr, owner);
    cu.doff = 0;
    cu.dcount = 0;
    cu.llk = bcast_f64(c.llk, owner);
    cu.g = bcast_words<KT>(c.g, owner);
    cu.memo_on = memo_mut;
    cu.mvalid = false;
    cu.gen = 1;
    cu.memo_gen = 1;
    // the unit's dictionary
    const int unit = __builtin_amdgcn_readlane(u, owner);
    if (unit != cur_dict_unit) {
      lds_sync();
      load_dict(LL, P.dict + (size_t)unit * DICT_MAX, (int)LL.ndict[cs], lane);
      cur_dict_unit = unit;
      lds_sync();
    }
  };

  // ---- initial likelihood (assemble/mcmc.py:303): one evaluation per chain, the wave serving chain after chain ----
  {
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
      const double v = spec_coop_all<KT, 64>(1ull, S.pw, S.shift, S.cols, S.nreads, S.ndict, S.dict, S.gptr, S.bw, false, S.crow, mmax, cu.Mh,
                                             (1u << cu.bits) - 1u, rpad, lane);
      lds_sync();
      const double l0 = __shfl(v, 0, WAVE);
      if (ci == cs) c.llk = l0;
    }
  }

  for (int step = 0; step < Sn; step++) {
    if (c.alive && isnan(c.llk)) {  // assemble/mcmc.py:330-331
      status = MCHAP_UNIT_NAN_LLK;
      c.alive = false;
    }
    bool changed = false;  // genotype changed during this step
    // =============================== mutation compound step ===============================
    {
      const int n = KT * Mh;
      const uint64_t ctr0 = c.ctr;
      const uint64_t base = ctr0 + (uint64_t)(n - 1);  // first uniform
      const int E = lane_extra(Mh);
      LDSP(uint64_t) win = LL.win + (size_t)ci * LL.win_n;
      bool ok = c.alive && c.mvalid;
      if (ok) {
        // the n uniforms (tested in registers) and the E draws behind them (the structural steps' window), Philox blocks
        // dealt round-robin to the chain's lanes
        const Stream st = ld_stream_l(LL.gstream, ci);
        const uint64_t b0 = base >> 1;
        const int nblk = (int)(((base + (uint64_t)(n + E) + 1) >> 1) - b0);
        for (int b = sl; b < nblk; b += L) {
          uint32_t o[4];
          const uint64_t blk = b0 + (uint64_t)b;
          philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), st.c2, st.c3, st.k0, st.k1, o);
          const long long i0 = (long long)(blk << 1) - (long long)base;
          const uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32), w1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
          if (i0 >= 0) {
            if (i0 < n) {
              const uint64_t x = u53_of(w0);
              ok = ok && x >= c.lo && x < c.hi;
            } else if (i0 < n + E) {
              win[i0 - n] = w0;
            }
          }
          if (i0 + 1 < n) {
            const uint64_t x = u53_of(w1);
            ok = ok && x >= c.lo && x < c.hi;
          } else if (i0 + 1 < n + E) {
            win[i0 + 1 - n] = w1;
          }
        }
      }
      lds_sync();  // the window entries were written by the chain's other lanes
      // all lanes of the chain must agree
      {
        const unsigned long long badm = __ballot(c.alive && !ok);
        const unsigned long long mine = L == 64 ? ~0ull : (((1ull << L) - 1ull) << (ci * L));
        ok = c.alive && !(badm & mine);
      }
      if (c.alive && ok) {
        c.ctr = ctr0 + (uint64_t)(2 * n - 1);
        c.doff = 0;
        c.dcount = E;
      }
      // chains that could not decide: the whole wave runs the step for them, one chain at a time
      unsigned long long todo = __ballot(c.alive && !ok && sl == 0);
      while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cs = owner >> lsh;
        Grp<KT> cu;
        SpecLds S;
        open_chain(cs, cu, S);
        const GWords<KT> before = cu.g;
        spec_mutation<KT, 64>(cu, S, 1.0, amax, mmax, nmax, rpad, lane, 0, lane);
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
            c.n_unknown = 2 * spec_memo_entries(Mh);
          }
        }
        if (diff)  // the interval-step thresholds described the previous genotype
          for (int i = lane; i < 2 * LL.tri; i += WAVE) LL.memo[(size_t)cs * 2 * LL.tri + i] = MEMO_UNKNOWN;
        lds_sync();
      }
    }
    // =============================== structural compound steps ===============================
#pragma unroll 1
    for (int kind = 0; kind < 3; kind++) {
      const int step_type = kind == 0 ? 0 : 1;
      LDSP(uint64_t) win = LL.win + (size_t)ci * LL.win_n;
      LDSP(uint32_t) memo = LL.memo + (size_t)ci * 2 * LL.tri + (size_t)step_type * LL.tri;
      // decision threshold: rand() <= p  <=>  u53 < pt
      const double pstep = kind == 0 ? D.p_recomb : (kind == 1 ? D.p_partial : D.p_dosage);
      const uint64_t pthr = pstep < 0.0 ? 0ull : (pstep >= 1.0 ? (1ull << 53) : (uint64_t)floor(pstep * 9007199254740992.0) + 1ull);
      bool pending = c.alive;   // chain has not finished this compound step
      bool exact = false;       // fast path could not decide: visiting order + walk
      uint64_t zeros = 0;
      int n_int = 0;
      int doff1 = 0;            // window offset after decision / breaks (start of the permutation draws)
      // ---- fast path; re-run after a refill for the chains whose window ran out ----
      while (wave_any(pending && !exact)) {
        bool short_ = false;
        const bool act = pending && !exact;
        int doff = c.doff;
        auto fetch = [&](int i) -> uint64_t {
          if (i < c.dcount) return win[i];
          short_ = true;
          return 0ull;
        };
        bool doit = false, bad_breaks = false;
        uint64_t zeros_ = 0;
        int n_int_ = 0;
        if (act) {
          doit = u53_of(fetch(doff)) < pthr;
          doff++;
          if (doit && kind < 2) {
            int nb;
            if (D.n_intervals > 0) {
              doff++;  // break_dist = [0, ..., 0, 1]: the draw is consumed (assemble/mcmc.py:214-217)
              nb = D.n_intervals - 1;
            } else {
              const uint64_t x = u53_of(fetch(doff));
              doff++;
              LDSP(uint64_t) bc = LL.bcum + (size_t)ci * mmax;
              nb = Mh;
              for (int j = 0; j < Mh; j++)
                if (x < bc[j]) {
                  nb = j;
                  break;
                }
            }
            if (nb >= Mh) {
              bad_breaks = true;
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
              zeros_ = ~ind & ((1ull << (Mh + 1)) - 1ull);
              n_int_ = nb + 1;
            }
          } else if (doit) {
            zeros_ = 1ull | (1ull << Mh);
            n_int_ = 1;
          }
        }
        const int doff1_ = doff;
        bool done = act && !doit;
        if (act && doit) {
          // order-free check: every interval known, and every consumed uniform beyond the largest threshold
          bool unknown = !memo_iv;
          int n_cons = 0;
          uint32_t mx = 0;
          uint64_t z = zeros_;
          for (int qq = 0; qq < n_int_; qq++) {
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
            for (int k = 0; k < n_cons; k++) low = low || (((uint32_t)fetch(doff + (n_int_ - 1) + k)) >> 5) < mx;
            if (!low) {
              doff += n_int_ - 1 + n_cons;
              done = true;
            }
          }
        }
        if (act && !short_) {
          if (bad_breaks) {
            status = MCHAP_UNIT_BREAKS;
            c.alive = false;
            pending = false;
          } else if (done) {
            c.ctr += (uint64_t)(doff - c.doff);
            c.doff = doff;
            pending = false;
          } else {
            exact = true;  // keeps c.ctr / c.doff at the start of the step; doff1 marks the permutation draws
            zeros = zeros_;
            n_int = n_int_;
            doff1 = doff1_;
          }
        }
        // refill the windows that ran out (from the chain's current draw) and go again
        if (wave_any(act && short_)) {
          const bool mine = act && short_;
          if (mine) {
            const Stream st = ld_stream_l(LL.gstream, ci);
            stage_draws_l(st, c.ctr, LL.win_n, win, sl, L);
          }
          lds_sync();
          if (mine) {
            c.doff = 0;
            c.dcount = LL.win_n;
          }
        }
      }
      // ---- exact path: visiting order (np.random.permutation), then the intervals one after the other ----
      if (wave_any(exact)) {
        LDSP(uint8_t) ord = LL.ord + (size_t)ci * (mmax + 1);
        // the window must hold the rest of the step: n_int - 1 shuffle draws + up to n_int uniforms
        if (wave_any(exact && c.dcount - doff1 < 2 * n_int)) {
          const bool mine = exact && c.dcount - doff1 < 2 * n_int;
          if (mine) {
            const Stream st = ld_stream_l(LL.gstream, ci);
            stage_draws_l(st, c.ctr, LL.win_n, win, sl, L);
          }
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
            if (ci == cs) {
              if (R.n_opt_req > 0) doff++;  // the step's uniform (structural.py:504-506: none without options)
              if (R.moved) {
                c.g = cu.g;
                c.llk = cu.llk;
                c.mvalid = false;
                changed = true;
                c.n_unknown = 2 * spec_memo_entries(Mh);
              } else {
                c.n_unknown -= R.filled;
                c.cursor += WAVE - 1;
                if (c.cursor >= 2 * spec_memo_entries(Mh)) c.cursor %= 2 * spec_memo_entries(Mh);
              }
              qi++;
            }
            if (R.moved) {  // the thresholds described the previous genotype
              for (int i = lane; i < 2 * LL.tri; i += WAVE) LL.memo[(size_t)cs * 2 * LL.tri + i] = MEMO_UNKNOWN;
              lds_sync();
            }
          }
        }
      }
    }
    // =============================== thresholds of settled chains ===============================
    if (memo_iv && wave_any(c.alive && c.n_unknown > 0 && !changed && c.stable)) {
      unsigned long long todo = __ballot(c.alive && c.n_unknown > 0 && !changed && c.stable && sl == 0);
      while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cs = owner >> lsh;
        Grp<KT> cu;
        SpecLds S;
        open_chain(cs, cu, S);
        const int cur = __builtin_amdgcn_readlane(c.cursor, owner);
        const ServeResult<KT> R = serve_structural<KT>(cu, S, LL, cs, false, 0, 0, 0ull, cur, mmax, rpad, lane);
        if (ci == cs) {
          c.n_unknown -= R.filled;
          c.cursor += WAVE;
          if (c.cursor >= 2 * spec_memo_entries(Mh)) c.cursor %= 2 * spec_memo_entries(Mh);
        }
      }
    }
    c.stable = !changed;
    // =============================== record ===============================
    {
      LDSP(uint64_t) tb = LL.tbuf + (size_t)ci * LANE_TB * (KT + 1);
      const int slot = step % LANE_TB;
      if (c.alive) {
        const GWords<KT> gr = c.g;
        for (int w = sl; w < KT; w += L) {
          const uint64_t x = sel_word<KT>(gr, w);
          int rank = 0;
#pragma unroll
          for (int h = 0; h < KT; h++) rank += (gr.w[h] < x || (gr.w[h] == x && h < w)) ? 1 : 0;
          tb[slot * KT + rank] = x;
        }
        if (sl == 0) tb[LANE_TB * KT + slot] = (uint64_t)__double_as_longlong(c.llk);
      }
      if (slot == LANE_TB - 1 || step == Sn - 1) {
        lds_sync();
        if (c.alive) {
          const int first = step - slot;
          const int nw = (slot + 1) * KT;
          uint64_t *tp = reinterpret_cast<uint64_t *>((uintptr_t)LL.gptr[ci * GP_N + GP_TRACE]) + (size_t)first * KT;
          for (int i = sl; i < nw; i += L) tp[i] = tb[i];
          uint64_t *lp = reinterpret_cast<uint64_t *>((uintptr_t)LL.gptr[ci * GP_N + GP_LLK]) + first;
          for (int i = sl; i <= slot; i += L) lp[i] = tb[LANE_TB * KT + i];
        }
        lds_sync();
      }
    }
  }
  if (status != MCHAP_UNIT_OK && sl == 0) atomicMax(&D.status[u], status);
}

}  // namespace mchap
