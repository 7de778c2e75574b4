// Table completion of the phased de-novo sampler (kernel 5) for MI355X (gfx950): ONE LANE PER REQUEST.
//
// Before a settled chain can coast (denovo_coast_kernel.hpp) the total move probability of every interval step
// (step type, start, stop) of its current genotype must be known (structural.py:433-673: the options of an interval,
// their Metropolis-Hastings ratios, the sum of their probabilities).  denovo_spec_kernel<.., true> completes that table
// with the code of a visit: 64 option slots per round, every distinct request of a round evaluated one after the other
// by the whole wavefront, lanes over reads -- a few microseconds of dependent latency each, ~90 of them per chain at
// BASELINE configs[1] and ~2300 at configs[4], where it is half of the sampler's time.
//
// The completion knows all its requests up front, so this kernel turns the evaluation around:
//   1. a chunk of still unknown intervals is listed, their options enumerated (the reference's order, the code of
//      denovo_spec_kernel) and their proposal genotypes de-duplicated through an LDS hash table (options of different
//      intervals coincide: ~650 option slots are ~90 distinct genotypes at configs[1]);
//   2. the distinct requests are evaluated 64 at a time, one per lane: a tile of the unit's float64 read table lives in
//      LDS as [row = (position, allele)][read], every lane walks the same reads and picks the factors of ITS alleles
//      (two candidate addresses per position: a broadcast read), multiplies them in position order, adds the haplotype
//      products in haplotype order (unchanged haplotypes take the chain's base products, also in LDS), takes read_log and
//      accumulates.  The reads are walked in the order of the wavefront butterfly -- lane-major, lanes in bit-reversed
//      order, a binary counter of partial sums -- so the sum is associated exactly as wave_sum() associates it:
//      the value is bit-identical to the wave-wide evaluation (same factors, same products, same sums, same tree);
//      padding reads (weight 0, term +-0.0) are skipped, which is an exact identity on every reachable sum;
//   3. the options' probabilities and the intervals' totals are formed as a visit without a move would form them.
// Independent lanes, no cross-lane traffic in the inner loop: ~80 VALU instructions per (request, read) instead of a
// latency-bound butterfly per request.  Results are those of the in-kernel completion bit for bit (tests/test_gpu_fill.py
// compares the tables); the sampler's traces do not change.
//
// Launch: one wavefront per chain of the list (PipeState records written by the exporting launch, which then skips its
// own completion: PIPE_NOFILL); dynamic LDS fill_lds_bytes().
#pragma once
#include "denovo_spec_kernel.hpp"

namespace mchap {

constexpr int FILL_HASH = 1024;   // open-addressing slots of a chunk's request table
constexpr int FILL_IV_CAP = 192;  // listed intervals waiting for a chunk
constexpr int FILL_BG = 4;        // batches (of 64 requests) evaluated per staging of a tile
// Option slots and distinct requests of one chunk of intervals.  Packed requests (one word each) are cheap to keep: a
// chunk then holds every interval of both step types at BASELINE configs[1] (648 slots); wide ones K words each.
__host__ __device__ inline int fill_slot_cap(int kw) { return kw == 1 ? 768 : 256; }
__host__ __device__ inline int fill_uniq_cap(int kw) { return 256; }

// Tile geometry: a tile holds the reads of `lt` lanes (lane positions in bit-reversed order) -- lt * nch slots -- as rows of
// (slots + 1) doubles.  Returns the largest lt in {64, 32, .., 1} that fits the budget, 0 if none does.
__host__ __device__ inline int fill_tile_lanes(int K, int max_pos, int max_allele, int rpad, size_t budget) {
  const int nch = rpad / 64;
  for (int lt = 64; lt >= 1; lt >>= 1) {
    const size_t slots = (size_t)lt * nch;
    const size_t bytes = ((size_t)max_pos * max_allele * (slots + 1) + (size_t)2 * K * slots + slots) * 8;
    if (bytes <= budget) return lt;
  }
  return 0;
}
// words kept per distinct request: the packed genotype when it fits one word, else its K haplotype words
__host__ __device__ inline int fill_key_words(int K, int max_pos, int max_allele) {
  return K * allele_bits(max_allele) * max_pos <= 64 ? 1 : K;
}
struct FillLds {
  size_t tile, bp, cw, stk, xst, pt, ln, lninv, cols, shift, ivse, ivlin, ivlout, ivno, ivoff, sliv, slopt, sluid, ukey, ullk, htab, total;
};
// (ptab, the options' probabilities, reuses the request words' array: those are no longer needed when it is written)
__host__ __device__ inline FillLds fill_lds(int K, int max_pos, int max_allele, int rpad, int lt) {
  FillLds L;
  const int kw = fill_key_words(K, max_pos, max_allele);
  const size_t slots = (size_t)(lt > 0 ? lt : 1) * (rpad / 64);
  const size_t nsl = (size_t)fill_slot_cap(kw), nuq = (size_t)fill_uniq_cap(kw);
  size_t o = 0;
  L.tile = o; o += (size_t)max_pos * max_allele * (slots + 1) * 8;
  L.bp = o; o += (size_t)2 * K * slots * 8;  // base terms bp[h] / K and their running sums in haplotype order
  L.cw = o; o += slots * 8;
  L.stk = o; o += (size_t)2 * 7 * 64 * 8;  // in-tile partial sums of the two requests a lane evaluates at a time
  {
    int ltl = 0;
    while ((1 << ltl) < (lt > 0 ? lt : 1)) ltl++;
    L.xst = o; o += (size_t)FILL_BG * (7 - ltl) * 64 * 8;  // a batch's partial sums above the tile level
  }
  L.pt = o; o += (size_t)(2 * K + 5) * 8;
  L.ln = o; o += (size_t)SPEC_LN * 8;
  L.lninv = o; o += (size_t)SPEC_LN * 8;
  { const size_t a = nuq * kw * 8, b = nsl * 8; L.ukey = o; o += a > b ? a : b; }
  L.ullk = o; o += nuq * 8;
  L.ivse = o; o += (size_t)FILL_IV_CAP * 4;
  L.ivlin = o; o += (size_t)FILL_IV_CAP * 4;
  L.ivlout = o; o += (size_t)FILL_IV_CAP * 4;
  L.ivno = o; o += (size_t)FILL_IV_CAP * 2;
  L.ivoff = o; o += (size_t)FILL_IV_CAP * 2;
  L.cols = o; o += (size_t)2 * max_pos;
  o = (o + 1) & ~(size_t)1;
  L.sluid = o; o += nsl * 2;
  L.htab = o; o += (size_t)FILL_HASH * 2;
  L.sliv = o; o += nsl;
  L.slopt = o; o += nsl;
  L.shift = o; o += (size_t)max_pos;
  L.total = (o + 63) & ~(size_t)63;
  return L;
}

// Launch geometry: lanes per tile (0: the shape does not fit -- the in-kernel completion serves it) and the kernel's dynamic LDS.
// Four waves per CU when a useful tile fits beside the fixed arrays, else two (bigger tiles: fewer stagings).
__host__ inline int fill_geometry(int K, int max_pos, int max_allele, int rpad, size_t *lds_bytes) {
  if (K < 2 || K > 8 || K * (K - 1) > 64) return 0;
  auto best = [&](size_t per_wave) {
    for (int lt = 64; lt >= 1; lt >>= 1)
      if (fill_lds(K, max_pos, max_allele, rpad, lt).total <= per_wave) return lt;
    return 0;
  };
  const int lt4 = best(40 * 1024), lt2 = best(80 * 1024);
  const int lt = (lt4 >= 8 || lt2 <= lt4) ? lt4 : lt2;
  if (lt > 0 && lds_bytes) *lds_bytes = fill_lds(K, max_pos, max_allele, rpad, lt).total;
  return lt;
}

__device__ __forceinline__ int brev6(int x) { return (int)(__brev((unsigned)x) >> 26); }

// allele (bits wide, at bit `sh`: wave-uniform) of a haplotype word held as two 32-bit halves: one v_bfe_u32 unless the field
// straddles the halves (3-bit alleles only)
__device__ __forceinline__ uint32_t fill_allele(uint32_t lo, uint32_t hi, int sh, int bits, uint32_t amask) {
  if (sh >= 32) return (hi >> (sh - 32)) & amask;
  if (sh + bits <= 32) return (lo >> sh) & amask;
  return __builtin_amdgcn_alignbit(hi, lo, sh) & amask;
}

// (start, stop) of entry e of an interval table: e = stop (stop - 1) / 2 + start, 0 <= start < stop
__device__ __forceinline__ void fill_entry_interval(int e, int &start, int &stop) {
  int s = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
  while (s * (s - 1) / 2 > e) s--;
  while ((s + 1) * s / 2 <= e) s++;
  stop = s;
  start = e - s * (s - 1) / 2;
}

// The my_o-th option of interval labels (lin, lout) in the reference's enumeration order (structural.py:121-178 /
// 240-307; the code of denovo_spec_kernel's spec_structural): the `in` label pack after the move.
template <int KT>
__device__ __forceinline__ uint32_t fill_option(int step_type, uint32_t lin, uint32_t lout, int my_o) {
  const uint32_t hd = dosage_of_labels(lin, lout, KT, true);
  uint32_t oin = 0;
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
  return oin;
}

template <int KT>
__global__ __launch_bounds__(64, 1) void denovo_fill_kernel(const SimtParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int lane = threadIdx.x;
  const int Cn = D.chains, Sn = D.steps, mmax = P.max_pos;
  const int E = spec_memo_entries(mmax);
  const long long n_chains = (long long)P.n_units * Cn;
  const int n_list = P.pipe_count ? *P.pipe_count : (int)n_chains;
  if ((long long)blockIdx.x >= n_list) return;  // the grid is sized for every chain
  const long long q = P.pipe_list ? (long long)P.pipe_list[blockIdx.x] : (long long)blockIdx.x;
  const PipeState *st = reinterpret_cast<const PipeState *>(P.pipe_state) + q;
  // (a chain that is finished, stopped by an error, or not settled -- it moved after its last full mutation step: the
  // coasting kernel hands it straight back -- does not get its tables completed: as in denovo_spec_kernel)
  if (st->step >= Sn || st->mvalid == 0) return;
  const int u = (int)(q / Cn);
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);
  const int Mh = mi[META_I_MH];
  const bool w01 = mi[META_I_W01] != 0;
  const int A = U.max_allele;
  const int bits = allele_bits(A);
  const uint32_t amask = (1u << bits) - 1u;
  const int R = U.n_reads;
  const int rpad = D.rpad, nch = rpad / WAVE;
  const int LT = P.fill_lt, n_tiles = WAVE / LT, slots = LT * nch, RS = slots + 1;
  int LTL = 0;
  while ((1 << LTL) < LT) LTL++;
  const int KW = P.fill_kw;
  const int SLOT_CAP = fill_slot_cap(KW), UNIQ_CAP = fill_uniq_cap(KW);
  const int key_bits = bits * Mh;
  const uint64_t key_mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull);
  const double inbreeding = U.inbreeding;
  const double invK = 1.0 / (double)KT;
  const double temp = D.temps[0];
  const FillLds L = fill_lds(KT, P.max_pos, P.max_allele, rpad, LT);
  LDSP(double) Tt = lds_cast<double>(smem + L.tile);
  LDSP(double) bp = lds_cast<double>(smem + L.bp);
  LDSP(double) cwt = lds_cast<double>(smem + L.cw);
  LDSP(double) stk = lds_cast<double>(smem + L.stk);
  LDSP(double) xst = lds_cast<double>(smem + L.xst);
  LDSP(double) pt = lds_cast<double>(smem + L.pt);
  LDSP(double) ln = lds_cast<double>(smem + L.ln);
  LDSP(double) lninv = lds_cast<double>(smem + L.lninv);
  LDSP(uint64_t) ukey = lds_cast<uint64_t>(smem + L.ukey);
  LDSP(double) ptab = lds_cast<double>(smem + L.ukey);  // (after the evaluation: see fill_lds)
  LDSP(double) ullk = lds_cast<double>(smem + L.ullk);
  LDSP(uint32_t) ivse = lds_cast<uint32_t>(smem + L.ivse);
  LDSP(uint32_t) ivlin = lds_cast<uint32_t>(smem + L.ivlin);
  LDSP(uint32_t) ivlout = lds_cast<uint32_t>(smem + L.ivlout);
  LDSP(uint16_t) ivno = lds_cast<uint16_t>(smem + L.ivno);
  LDSP(uint16_t) ivoff = lds_cast<uint16_t>(smem + L.ivoff);
  LDSP(uint16_t) cols = lds_cast<uint16_t>(smem + L.cols);
  LDSP(uint16_t) sluid = lds_cast<uint16_t>(smem + L.sluid);
  LDSP(uint16_t) htab = lds_cast<uint16_t>(smem + L.htab);
  LDSP(uint8_t) sliv = lds_cast<uint8_t>(smem + L.sliv);
  LDSP(uint8_t) slopt = lds_cast<uint8_t>(smem + L.slopt);
  LDSP(uint8_t) shift = lds_cast<uint8_t>(smem + L.shift);

  for (int i = lane; i < SPEC_LN; i += WAVE) {
    ln[i] = c_ln[i];
    lninv[i] = c_ln_inv[i];
  }
  for (int j = lane; j < Mh; j += WAVE) {
    cols[j] = (uint16_t)mi[META_I_COLS + j];
    shift[j] = (uint8_t)(bits * (Mh - 1 - j));
  }
  if (!isnan(inbreeding))
    for (int i = lane; i < 2 * KT + 5; i += WAVE) pt[i] = mf[meta_f_prior(0) + i];
  GWords<KT> g;  // the chain's current genotype, in the chain's own haplotype order
#pragma unroll
  for (int h = 0; h < KT; h++) g.w[h] = st->g[h];
  const double cur_llk = st->llk;
  GLBP(const double) rt = (GLBP(const double))(P.rt + (size_t)u * P.max_ma * rpad);
  GLBP(const double) cwg = (GLBP(const double))(P.cntw + (size_t)u * rpad);
  double *memo = P.pipe_memo + (size_t)q * 2 * E;
  const uint64_t full = mask_of(bits, Mh, 0, Mh);
  const int n_entries = spec_memo_entries(Mh);
  lds_sync();
  const double lprior_cur = isnan(inbreeding) ? 0.0 : prior_of<KT>(pt, inbreeding, dosage_words<KT>(g));

  // Stage tile t: the reads of lanes l with brev6(l) in [t LT, (t + 1) LT) -- slot (brev6(l) - t LT) nch + i holds read
  // l + 64 i -- as T[(j, a)][slot], the read weights, and the base products bp[h][slot] of the current genotype (factors in
  // position order from 1.0, as spec_hap_prod forms them).  Every lane copies elements (row, slot) of the tile, eight loads
  // in flight at a time.
  const int MA = Mh * A;
  LDSP(double) pre = bp + KT * slots;  // pre[h][slot] = (((0 + bpc[0]) + bpc[1]) + ..) + bpc[h - 1]: the sum a request whose
                                       // first changed haplotype is h starts from
  auto stage = [&](int t) {
    // the lane's slots s = lane + 64 k and the reads they hold
    constexpr int SK = 8;  // (slots <= 512: fill_geometry's budgets)
    int rr[SK];
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int s_ = lane + WAVE * k;
      const int pl = s_ / nch, i = s_ - pl * nch;
      rr[k] = s_ < slots ? brev6(t * LT + pl) + WAVE * i : -1;
    }
    for (int row0 = 0; row0 < MA; row0 += 8) {
#pragma unroll
      for (int k = 0; k < SK; k++) {
        if (WAVE * k >= slots) break;
        double v[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
          const int row = min(row0 + x, MA - 1);
          const int j = row / A, a = row - j * A;
          v[x] = rr[k] >= 0 ? rt[(size_t)((int)cols[j] + a) * rpad + rr[k]] : 0.0;
        }
#pragma unroll
        for (int x = 0; x < 8; x++)
          if (row0 + x < MA && rr[k] >= 0) Tt[(row0 + x) * RS + lane + WAVE * k] = v[x];
      }
    }
#pragma unroll
    for (int k = 0; k < SK; k++)
      if (rr[k] >= 0) cwt[lane + WAVE * k] = cwg[rr[k]];
    lds_sync();
    for (int s_ = lane; s_ < slots; s_ += WAVE) {
      double run = 0.0;
#pragma unroll
      for (int h = 0; h < KT; h++) {
        const uint64_t wh = g.w[h];
        double pr = 1.0;
        for (int j = 0; j < Mh; j++) {
          const uint32_t a = (uint32_t)(wh >> (bits * (Mh - 1 - j))) & amask;
          pr *= Tt[(j * A + (int)a) * RS + s_];
        }
        const double term = pr * invK;
        bp[h * slots + s_] = term;
        pre[h * slots + s_] = run;
        run += term;
      }
    }
    lds_sync();
  };
  int staged = -1;
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
  unsigned long long fph[12];
  for (int i_ = 0; i_ < 12; i_++) fph[i_] = 0;
  unsigned long long ft0 = __builtin_amdgcn_s_memtime();
#define FPH(i)                                                    \
  do {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
    fph[i] += t_ - ft0;                                           \
    ft0 = t_;                                                     \
  } while (0)
#define FCNT(i, n) fph[i] += (n)
#else
#define FPH(i)
#define FCNT(i, n)
#endif

  // (step type, entry) pairs are walked as one sequence ge = step_type * n_entries + e; unknown entries with options
  // wait in the interval list [iv0, n_iv) for a chunk
  int ge_cursor = 0, iv0 = 0, n_iv = 0;
  while (true) {
    // ---- (1) list unknown intervals: labels, option counts ----
    if (iv0 == n_iv) iv0 = n_iv = 0;
    while (n_iv + WAVE <= FILL_IV_CAP && ge_cursor < 2 * n_entries) {
      const int ge = ge_cursor + lane;
      const int ty = ge >= n_entries ? 1 : 0;
      const int e = ge - ty * n_entries;
      const bool in = ge < 2 * n_entries;
      double *mt = memo + (size_t)ty * E;
      bool unk = in && isnan(mt[e]);
      uint32_t lin = 0, lout = 0;
      int no = 0, start = 0, stop = 1;
      if (unk) {
        fill_entry_interval(e, start, stop);
        const uint64_t min_ = mask_of(bits, Mh, start, stop);
        lin = seg_labels<KT>(g, min_);
        lout = seg_labels<KT>(g, full & ~min_);
        no = ty == 0 ? recombination_n_options(lin, lout, KT) : dosage_n_options(lin, lout, KT);
        if (no == 0) {
          mt[e] = -1.0;  // no options: the step consumes no draw
          unk = false;
        }
      }
      const unsigned long long um = __ballot(unk);
      if (unk) {
        const int k = n_iv + __popcll(um & ((1ull << lane) - 1ull));
        ivse[k] = (uint32_t)start | ((uint32_t)stop << 8) | ((uint32_t)e << 16) | ((uint32_t)ty << 31);
        ivlin[k] = lin;
        ivlout[k] = lout;
        ivno[k] = (uint16_t)no;
      }
      n_iv += __popcll(um);
      ge_cursor += WAVE;
    }
    lds_sync();
    FPH(0);
    if (iv0 == n_iv) break;  // every entry of both tables is known

    // ---- (2) a chunk: rounds of whole intervals (<= 64 option slots each); the slots' proposal genotypes are
    //      de-duplicated through the hash table: slot -> index of its distinct request ----
    for (int i = lane; i < FILL_HASH; i += WAVE) htab[i] = 0;
    lds_sync();
    const int chunk_iv0 = iv0;
    int n_slots = 0, n_uniq = 0;
    auto slot_request = [&](int s, GWords<KT> &pw, uint32_t &oin, uint32_t &lo, int &my_no, int &ty) {
      const int ii = chunk_iv0 + (int)sliv[s];
      const uint32_t se = ivse[ii];
      ty = (int)(se >> 31);
      const uint64_t min_ = mask_of(bits, Mh, (int)(se & 255u), (int)((se >> 8) & 255u));
      const uint32_t li = ivlin[ii];
      lo = ivlout[ii];
      my_no = ivno[ii];
      oin = fill_option<KT>(ty, li, lo, (int)slopt[s]);
#pragma unroll
      for (int h = 0; h < KT; h++) pw.w[h] = (g.w[h] & ~min_) | (sel_word<KT>(g, (int)nib(oin, h)) & min_);
    };
    while (iv0 < n_iv && n_uniq + WAVE <= UNIQ_CAP && n_slots + WAVE <= SLOT_CAP && iv0 - chunk_iv0 < 255) {
      // the round's intervals: as many as fit 64 slots; lane k < count takes interval iv0 + k
      int no_k = (iv0 + lane < n_iv) ? (int)ivno[iv0 + lane] : 1 << 20;
      int incl = no_k;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const int v = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl = min(incl + v, 1 << 20);
      }
      const unsigned long long fm = __ballot(incl <= WAVE && iv0 - chunk_iv0 + lane < 255);
      const int cnt = __popcll(fm);  // (a prefix: the counts are positive; at least one interval fits: <= 56 options)
      const int used = __shfl(incl, cnt - 1, WAVE);
      if (lane < cnt) {
        const int off = n_slots + incl - no_k;
        ivoff[iv0 + lane] = (uint16_t)off;
        for (int o = 0; o < no_k; o++) {
          sliv[off + o] = (uint8_t)(iv0 - chunk_iv0 + lane);
          slopt[off + o] = (uint8_t)o;
        }
      }
      lds_sync();
      const int s = n_slots + lane;
      const bool act = lane < used;
      GWords<KT> pw = g;
      if (act) {
        uint32_t oin, lo;
        int my_no, ty;
        slot_request(s, pw, oin, lo, my_no, ty);
      }
      uint64_t tag = 0;
      if (KW == 1) {
#pragma unroll
        for (int h = 0; h < KT; h++) tag = (tag << key_bits) | pw.w[h];
      } else {
#pragma unroll
        for (int h = 0; h < KT; h++) tag = mix64(tag ^ pw.w[h]) + 0x9E3779B97F4A7C15ull;
      }
      uint32_t hsh = (uint32_t)tag ^ ((uint32_t)(tag >> 32) * 0x9E3779B1u);
      hsh ^= hsh >> 16;
      hsh *= 0x7FEB352Du;
      hsh ^= hsh >> 15;
      int found = -1;
      if (act) {
        for (int pr = 0; pr < FILL_HASH; pr++) {
          const int e2 = htab[(hsh + pr) & (FILL_HASH - 1)];
          if (e2 == 0) break;
          bool same;
          if (KW == 1) {
            same = ukey[e2 - 1] == tag;
          } else {
            same = true;
#pragma unroll
            for (int h = 0; h < KT; h++) same = same && ukey[(e2 - 1) * KT + h] == pw.w[h];
          }
          if (same) {
            found = e2 - 1;
            break;
          }
        }
      }
      unsigned long long pend = __ballot(act && found < 0);
      while (pend) {
        const int l0 = __ffsll((long long)pend) - 1;
        bool same = act && found < 0;
        if (KW == 1) {
          same = same && tag == __shfl(tag, l0, WAVE);
        } else {
#pragma unroll
          for (int h = 0; h < KT; h++) same = same && pw.w[h] == __shfl(pw.w[h], l0, WAVE);
        }
        if (same) found = n_uniq;
        if (lane == l0) {
          if (KW == 1) {
            ukey[n_uniq] = tag;
          } else {
#pragma unroll
            for (int h = 0; h < KT; h++) ukey[n_uniq * KT + h] = pw.w[h];
          }
          for (int pr = 0; pr < FILL_HASH; pr++) {
            const int hi = (hsh + pr) & (FILL_HASH - 1);
            if (htab[hi] == 0) {
              htab[hi] = (uint16_t)(n_uniq + 1);
              break;
            }
          }
        }
        n_uniq++;
        pend &= ~__ballot(same);
      }
      if (act) sluid[s] = (uint16_t)found;
      n_slots += used;
      iv0 += cnt;
      lds_sync();
    }
    const int chunk_ivs = iv0 - chunk_iv0;
    FPH(1);
    FCNT(6, 1);
    FCNT(7, n_uniq);
    FCNT(8, n_slots);

    // ---- (3) the distinct requests: one per lane and NQ = 2 batches at a time (two independent requests per lane keep
    //      the SIMD's pipeline busy where a single wavefront's dependent chain would not); FILL_BG batches share each
    //      staging of a tile ----
    constexpr int NQ = 2;
    for (int bg0 = 0; bg0 < n_uniq; bg0 += FILL_BG * WAVE) {
      for (int t = 0; t < n_tiles; t++) {
        if (staged != t) {
          stage(t);
          staged = t;
          FPH(2);
          FCNT(9, 1);
        }
        for (int bi = 0; bi < FILL_BG; bi += NQ) {
          if (bg0 + bi * WAVE >= n_uniq) break;
          // the lane's requests: which haplotypes differ from the current genotype (at most two: a dosage option replaces
          // one word, a recombination swaps segments of two), and their words
          int h1[NQ], h2[NQ];
          uint64_t w1[NQ], w2[NQ];
          bool valid[NQ];
          bool any2 = false;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int uid = bg0 + (bi + q) * WAVE + lane;
            valid[q] = uid < n_uniq;
            h1[q] = 0;
            h2[q] = -1;
            w1[q] = g.w[0];
            w2[q] = 0;
            if (valid[q]) {
              int nd = 0;
#pragma unroll
              for (int h = 0; h < KT; h++) {
                uint64_t wh;
                if (KW == 1) wh = (ukey[uid] >> (key_bits * (KT - 1 - h))) & key_mask;
                else wh = ukey[uid * KT + h];
                if (wh != g.w[h]) {
                  if (nd == 0) {
                    h1[q] = h;
                    w1[q] = wh;
                  } else {
                    h2[q] = h;
                    w2[q] = wh;
                  }
                  nd++;
                }
              }
            }
            any2 = any2 || h2[q] >= 0;
            if (h2[q] < 0) w2[q] = w1[q];
          }
          any2 = wave_any(any2);
          uint32_t w1lo[NQ], w1hi[NQ], w2lo[NQ], w2hi[NQ];
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            w1lo[q] = (uint32_t)w1[q];
            w1hi[q] = (uint32_t)(w1[q] >> 32);
            w2lo[q] = (uint32_t)w2[q];
            w2hi[q] = (uint32_t)(w2[q] >> 32);
          }
          for (int pl = 0; pl < LT; pl++) {
            const int p = t * LT + pl;
            const int l = brev6(p);
            const int n_l = l < R ? (R - l + WAVE - 1) / WAVE : 0;  // chunks of lane l that hold reads
            double s_l[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) s_l[q] = 0.0;
            for (int c0 = 0; c0 < n_l; c0 += 4) {
              const int nb = min(4, n_l - c0);
              const int sb = pl * nch + c0;  // the block's first slot (the tile has a spare column: x < 4 stays inside)
              double cwv[4];
#pragma unroll
              for (int x = 0; x < 4; x++) cwv[x] = cwt[sb + x];
              double p1[NQ][4], p2[NQ][4];
#pragma unroll
              for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int x = 0; x < 4; x++) {
                  p1[q][x] = 1.0;
                  p2[q][x] = 1.0;
                }
              // positions four at a time: all their loads are issued before the first multiply (the factors are multiplied
              // in position order, as spec_hap_prod does)
              for (int j0 = 0; j0 < Mh; j0 += 4) {
                const int nj = min(4, Mh - j0);  // wave-uniform
                double f1[NQ][4][4], f2[NQ][4][4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                  if (jj < nj) {
                    const int j = j0 + jj;
                    const int sh = bits * (Mh - 1 - j);
                    const int rb = j * A * RS + sb;
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                      const uint32_t a1 = fill_allele(w1lo[q], w1hi[q], sh, bits, amask);
                      LDSP(double) r1 = Tt + (int)a1 * RS + rb;
#pragma unroll
                      for (int x = 0; x < 4; x++) f1[q][jj][x] = r1[x];
                      if (any2) {
                        const uint32_t a2 = fill_allele(w2lo[q], w2hi[q], sh, bits, amask);
                        LDSP(double) r2 = Tt + (int)a2 * RS + rb;
#pragma unroll
                        for (int x = 0; x < 4; x++) f2[q][jj][x] = r2[x];
                      }
                    }
                  }
                }
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                  if (jj < nj) {
#pragma unroll
                    for (int q = 0; q < NQ; q++)
#pragma unroll
                      for (int x = 0; x < 4; x++) {
                        p1[q][x] *= f1[q][jj][x];
                        if (any2) p2[q][x] *= f2[q][jj][x];
                      }
                  }
                }
              }
              // acc = sum over the haplotypes, in haplotype order, of (product / K): the terms before the request's first
              // changed haplotype are the base genotype's -- their running sum is tabulated (pre) --, then the changed
              // haplotype's own term, then the rest one by one (adding +0.0 for the lanes already past is exact)
#pragma unroll
              for (int q = 0; q < NQ; q++) {
                double acc[4];
#pragma unroll
                for (int x = 0; x < 4; x++) acc[x] = pre[h1[q] * slots + sb + x] + p1[q][x] * invK;
#pragma unroll
                for (int h = 1; h < KT; h++) {
#pragma unroll
                  for (int x = 0; x < 4; x++) {
                    double term = bp[h * slots + sb + x];
                    if (any2) term = (h == h2[q]) ? p2[q][x] * invK : term;
                    acc[x] += (h > h1[q]) ? term : 0.0;
                  }
                }
                double blk = 0.0;
                if (w01) {  // one logarithm for the block (read_log_sum: what every other kernel forms for these reads)
                  double y[4];
#pragma unroll
                  for (int x = 0; x < 4; x++) y[x] = (x < nb && cwv[x] != 0.0) ? acc[x] : 1.0;
                  blk = read_log_product<4>(y);
                } else {
#pragma unroll
                  for (int x = 0; x < 4; x++)
                    if (x < nb) blk += read_log(acc[x]) * cwv[x];
                }
                s_l[q] += blk;
              }
            }
            // the leaf of lane l joins the butterfly's tree: a binary counter of partial sums (levels below the tile's
            // in `stk`, the batch's levels above it in `xst`)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              LDSP(double) xs = xst + (bi + q) * (7 - LTL) * WAVE;
              LDSP(double) sk = stk + q * 7 * WAVE;
              double v = s_l[q];
              int lvl = 0;
              for (int pp = p; pp & 1; pp >>= 1, lvl++) v = (lvl < LTL ? sk[lvl * WAVE + lane] : xs[(lvl - LTL) * WAVE + lane]) + v;
              if (lvl < LTL) sk[lvl * WAVE + lane] = v;
              else xs[(lvl - LTL) * WAVE + lane] = v;
            }
          }
          if (t == n_tiles - 1) {
#pragma unroll
            for (int q = 0; q < NQ; q++)
              if (valid[q]) ullk[bg0 + (bi + q) * WAVE + lane] = xst[((bi + q) * (7 - LTL) + (6 - LTL)) * WAVE + lane];
          }
          FPH(3);
          FCNT(10, 1);
        }
      }
    }
    lds_sync();

    // ---- (4) option probabilities, then the totals a visit without a move would have formed ----
    for (int base = 0; base < n_slots; base += WAVE) {
      const int s = base + lane;
      double pr = 0.0;
      if (s < n_slots) {
        GWords<KT> pw;
        uint32_t oin, lo;
        int my_no, ty;
        slot_request(s, pw, oin, lo, my_no, ty);
        const double llk_i = ullk[sluid[s]];
        double lprior_ratio = 0.0;
        if (!isnan(inbreeding)) lprior_ratio = prior_of<KT>(pt, inbreeding, dosage_of_labels(oin, lo, KT, true)) - lprior_cur;
        const int n_return = ty == 0 ? recombination_n_options(oin, lo, KT) : dosage_n_options(oin, lo, KT);
        const double lproposal_ratio = lninv[n_return] - lninv[my_no];
        const double mh = ((llk_i - cur_llk) + lprior_ratio) * temp + lproposal_ratio;
        pr = exp(fmin(0.0, mh) - ln[my_no]);
      }
      lds_sync();  // (ptab shares the request words' array: every lane has read its request before anything is written)
      if (s < n_slots) ptab[s] = pr;
    }
    lds_sync();
    for (int k = lane; k < chunk_ivs; k += WAVE) {
      const int ii = chunk_iv0 + k;
      const int no2 = ivno[ii], off2 = ivoff[ii];
      const uint32_t se = ivse[ii];
      double cacc = 0.0;
      for (int o = 0; o < no2; o++) cacc += ptab[off2 + o];
      memo[(size_t)(se >> 31) * E + ((se >> 16) & 0x7FFFu)] = cacc;
    }
    lds_sync();
    FPH(4);
  }
#if defined(MCHAP_STATS) || defined(MCHAP_PHASES)
  FCNT(11, 1);
  if (lane == 0)
    for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_stats[i_], fph[i_]);
#endif
}

}  // namespace mchap
