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

constexpr int FILL_SLOTS = 256;  // option slots of one chunk of intervals (>= K (K - 1) = 56: one interval always fits)
constexpr int FILL_HASH = 512;   // open-addressing slots of the chunk's request table
constexpr int FILL_TILE_BUDGET = 14 * 1024;  // bytes of LDS for a tile of the read table + base products + weights

// Tile geometry: a tile holds the reads of `lt` lanes (lane positions in bit-reversed order) -- lt * nch slots -- as rows of
// (slots + 1) doubles.  Returns the largest lt in {64, 32, .., 1} that fits the budget, 0 if none does.
__host__ __device__ inline int fill_tile_lanes(int K, int max_pos, int max_allele, int rpad) {
  const int nch = rpad / 64;
  for (int lt = 64; lt >= 1; lt >>= 1) {
    const size_t slots = (size_t)lt * nch;
    const size_t bytes = ((size_t)max_pos * max_allele * (slots + 1) + (size_t)K * slots + slots) * 8;
    if (bytes <= (size_t)FILL_TILE_BUDGET) return lt;
  }
  return 0;
}
// words kept per distinct request: the packed genotype when it fits one word, else its K haplotype words
__host__ __device__ inline int fill_key_words(int K, int max_pos, int max_allele) {
  return K * allele_bits(max_allele) * max_pos <= 64 ? 1 : K;
}
struct FillLds {
  size_t tile, bp, cw, stk, pt, ln, lninv, cols, shift, ivse, ivlin, ivlout, ivno, ivoff, sliv, slopt, sluid, ptab, ukey, ullk, htab, total;
};
__host__ __device__ inline FillLds fill_lds(int K, int max_pos, int max_allele, int rpad) {
  FillLds L;
  const int lt = fill_tile_lanes(K, max_pos, max_allele, rpad);
  const size_t slots = (size_t)(lt > 0 ? lt : 1) * (rpad / 64);
  size_t o = 0;
  L.tile = o; o += (size_t)max_pos * max_allele * (slots + 1) * 8;
  L.bp = o; o += (size_t)K * slots * 8;
  L.cw = o; o += slots * 8;
  L.stk = o; o += (size_t)7 * 64 * 8;
  L.pt = o; o += (size_t)(2 * K + 5) * 8;
  L.ln = o; o += (size_t)SPEC_LN * 8;
  L.lninv = o; o += (size_t)SPEC_LN * 8;
  L.ukey = o; o += (size_t)FILL_SLOTS * fill_key_words(K, max_pos, max_allele) * 8;
  L.ullk = o; o += (size_t)FILL_SLOTS * 8;
  L.ptab = o; o += (size_t)FILL_SLOTS * 8;
  L.ivse = o; o += 64 * 4;
  L.ivlin = o; o += 64 * 4;
  L.ivlout = o; o += 64 * 4;
  L.ivno = o; o += 64 * 2;
  L.ivoff = o; o += 64 * 2;
  L.cols = o; o += (size_t)2 * max_pos;
  o = (o + 1) & ~(size_t)1;
  L.sluid = o; o += (size_t)FILL_SLOTS * 2;
  L.htab = o; o += (size_t)FILL_HASH * 2;
  L.sliv = o; o += FILL_SLOTS;
  L.slopt = o; o += FILL_SLOTS;
  L.shift = o; o += (size_t)max_pos;
  L.total = (o + 63) & ~(size_t)63;
  return L;
}
__host__ __device__ inline size_t fill_lds_bytes(int K, int max_pos, int max_allele, int rpad) {
  return fill_lds(K, max_pos, max_allele, rpad).total;
}
// shapes the kernel takes: a tile of one lane's reads must fit (the in-kernel completion serves the others)
__host__ __device__ inline bool fill_supported(int K, int max_pos, int max_allele, int rpad) {
  return K >= 2 && K <= 8 && fill_tile_lanes(K, max_pos, max_allele, rpad) > 0 && K * (K - 1) <= FILL_SLOTS;
}

__device__ __forceinline__ int brev6(int x) { return (int)(__brev((unsigned)x) >> 26); }

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
__global__ __launch_bounds__(64, 2) void denovo_fill_kernel(const SimtParams P) {
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
  const int A = U.max_allele;
  const int bits = allele_bits(A);
  const uint32_t amask = (1u << bits) - 1u;
  const int R = U.n_reads;
  const int rpad = D.rpad, nch = rpad / WAVE;
  const int LT = P.fill_lt, n_tiles = WAVE / LT, slots = LT * nch, RS = slots + 1;
  const int KW = P.fill_kw;
  const int key_bits = bits * Mh;
  const double inbreeding = U.inbreeding;
  const double invK = 1.0 / (double)KT;
  const double temp = D.temps[0];
  const FillLds L = fill_lds(KT, P.max_pos, P.max_allele, rpad);
  LDSP(double) Tt = lds_cast<double>(smem + L.tile);
  LDSP(double) bp = lds_cast<double>(smem + L.bp);
  LDSP(double) cwt = lds_cast<double>(smem + L.cw);
  LDSP(double) stk = lds_cast<double>(smem + L.stk);
  LDSP(double) pt = lds_cast<double>(smem + L.pt);
  LDSP(double) ln = lds_cast<double>(smem + L.ln);
  LDSP(double) lninv = lds_cast<double>(smem + L.lninv);
  LDSP(uint64_t) ukey = lds_cast<uint64_t>(smem + L.ukey);
  LDSP(double) ullk = lds_cast<double>(smem + L.ullk);
  LDSP(double) ptab = lds_cast<double>(smem + L.ptab);
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
  const double *rt = P.rt + (size_t)u * P.max_ma * rpad;
  const double *cwg = P.cntw + (size_t)u * rpad;
  double *memo = P.pipe_memo + (size_t)q * 2 * E;
  const uint64_t full = mask_of(bits, Mh, 0, Mh);
  lds_sync();
  const double lprior_cur = isnan(inbreeding) ? 0.0 : prior_of<KT>(pt, inbreeding, dosage_words<KT>(g));

  // Stage tile t: the reads of lanes l with brev6(l) in [t LT, (t + 1) LT) -- slot (brev6(l) - t LT) nch + i holds read
  // l + 64 i -- as T[(j, a)][slot], the read weights, and the base products bp[h][slot] of the current genotype
  // (factors in position order from 1.0, as spec_hap_prod forms them).
  auto stage = [&](int t) {
    const int p = brev6(lane) - t * LT;  // this lane's position within the tile
    if (p >= 0 && p < LT) {
      for (int i = 0; i < nch; i++) {
        const int r = lane + WAVE * i, s = p * nch + i;
        cwt[s] = cwg[r];
        for (int j = 0; j < Mh; j++) {
          const int c0 = cols[j];
          for (int a = 0; a < A; a++) Tt[(size_t)(j * A + a) * RS + s] = rt[(size_t)(c0 + a) * rpad + r];
        }
      }
    }
    lds_sync();
    for (int s = lane; s < slots; s += WAVE) {
#pragma unroll
      for (int h = 0; h < KT; h++) {
        const uint64_t wh = g.w[h];
        double pr = 1.0;
        for (int j = 0; j < Mh; j++) {
          const uint32_t a = (uint32_t)(wh >> shift[j]) & amask;
          pr *= Tt[(size_t)(j * A + (int)a) * RS + s];
        }
        bp[h * slots + s] = pr;
      }
    }
    lds_sync();
  };
  int staged = -1;

  for (int step_type = 0; step_type < 2; step_type++) {
    double *mtot = memo + (size_t)step_type * E;
    const int n_entries = spec_memo_entries(Mh);
    int next_e = 0;
    while (next_e < n_entries) {
      // ---- (1) the next unknown intervals: labels, option counts, slot offsets ----
      const int e = next_e + lane;
      const bool unk = e < n_entries && isnan(mtot[e]);
      const unsigned long long um = __ballot(unk);
      const int rank = __popcll(um & ((1ull << lane) - 1ull));
      uint32_t lin = 0, lout = 0;
      int no = 0, start = 0, stop = 1;
      if (unk) {
        fill_entry_interval(e, start, stop);
        const uint64_t min_ = mask_of(bits, Mh, start, stop);
        lin = seg_labels<KT>(g, min_);
        lout = seg_labels<KT>(g, full & ~min_);
        no = step_type == 0 ? recombination_n_options(lin, lout, KT) : dosage_n_options(lin, lout, KT);
        if (no == 0) mtot[e] = -1.0;  // no options: the step consumes no draw
      }
      // exclusive prefix sum of the option counts over the unknown intervals (in entry order)
      int incl = no;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const int v = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += v;
      }
      const int off = incl - no;
      const bool fits = unk && incl <= FILL_SLOTS;
      const unsigned long long fm = __ballot(fits);
      const unsigned long long rest = um & ~fm;  // (a prefix fits: the counts are non-negative)
      const int n_iv = __popcll(fm);
      const int n_slots = __shfl(incl, fm ? 63 - __clzll(fm) : 0, WAVE) * (fm ? 1 : 0);
      next_e = rest ? next_e + (__ffsll((long long)rest) - 1) : next_e + WAVE;
      if (fits) {
        ivse[rank] = (uint32_t)start | ((uint32_t)stop << 8) | ((uint32_t)e << 16);
        ivlin[rank] = lin;
        ivlout[rank] = lout;
        ivno[rank] = (uint16_t)no;
        ivoff[rank] = (uint16_t)off;
        for (int o = 0; o < no; o++) {
          sliv[off + o] = (uint8_t)rank;
          slopt[off + o] = (uint8_t)o;
        }
      }
      for (int i = lane; i < FILL_HASH; i += WAVE) htab[i] = 0;
      lds_sync();
      if (n_slots == 0) continue;

      // ---- (2) the slots' proposal genotypes, de-duplicated: slot -> index of its distinct request ----
      auto slot_request = [&](int s, GWords<KT> &pw, uint32_t &oin, uint32_t &lo, int &my_no) {
        const int ii = sliv[s];
        const uint32_t se = ivse[ii];
        const uint64_t min_ = mask_of(bits, Mh, (int)(se & 255u), (int)((se >> 8) & 255u));
        const uint32_t li = ivlin[ii];
        lo = ivlout[ii];
        my_no = ivno[ii];
        oin = fill_option<KT>(step_type, li, lo, (int)slopt[s]);
#pragma unroll
        for (int h = 0; h < KT; h++) pw.w[h] = (g.w[h] & ~min_) | (sel_word<KT>(g, (int)nib(oin, h)) & min_);
      };
      int n_uniq = 0;
      for (int base = 0; base < n_slots; base += WAVE) {
        const int s = base + lane;
        const bool act = s < n_slots;
        GWords<KT> pw = g;
        if (act) {
          uint32_t oin, lo;
          int my_no;
          slot_request(s, pw, oin, lo, my_no);
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
              for (int h = 0; h < KT; h++) same = same && ukey[(size_t)(e2 - 1) * KT + h] == pw.w[h];
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
              for (int h = 0; h < KT; h++) ukey[(size_t)n_uniq * KT + h] = pw.w[h];
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
        lds_sync();
      }

      // ---- (3) the distinct requests, 64 at a time, one per lane ----
      for (int b0 = 0; b0 < n_uniq; b0 += WAVE) {
        const int uid = b0 + lane;
        const bool valid = uid < n_uniq;
        // the lane's request: which haplotypes differ from the current genotype (at most two: a dosage option replaces
        // one word, a recombination swaps segments of two), and their words
        int h1 = 0, h2 = -1;
        uint64_t w1 = g.w[0], w2 = 0;
        if (valid) {
          int nd = 0;
#pragma unroll
          for (int h = 0; h < KT; h++) {
            uint64_t wh;
            if (KW == 1) wh = (ukey[uid] >> (key_bits * (KT - 1 - h))) & (key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull));
            else wh = ukey[(size_t)uid * KT + h];
            if (wh != g.w[h]) {
              if (nd == 0) {
                h1 = h;
                w1 = wh;
              } else {
                h2 = h;
                w2 = wh;
              }
              nd++;
            }
          }
        }
        const bool any2 = wave_any(h2 >= 0);
        if (h2 < 0) w2 = w1;
        const uint32_t w1lo = (uint32_t)w1, w1hi = (uint32_t)(w1 >> 32), w2lo = (uint32_t)w2, w2hi = (uint32_t)(w2 >> 32);
        for (int t = 0; t < n_tiles; t++) {
          if (staged != t) {
            stage(t);
            staged = t;
          }
          for (int pl = 0; pl < LT; pl++) {
            const int p = t * LT + pl;
            const int l = brev6(p);
            const int n_l = l < R ? (R - l + WAVE - 1) / WAVE : 0;  // chunks of lane l that hold reads
            double s_l = 0.0;
            for (int c0 = 0; c0 < n_l; c0 += 4) {
              const int nb = min(4, n_l - c0);
              const int sb = pl * nch + c0;  // the block's first slot (its four slots exist: padding reads are staged)
              double p1[4], p2[4];
#pragma unroll
              for (int x = 0; x < 4; x++) {
                p1[x] = 1.0;
                p2[x] = 1.0;
              }
              for (int j = 0; j < Mh; j++) {
                const int sh = __builtin_amdgcn_readfirstlane((int)shift[j]);
                const uint32_t a1 = (sh >= 32 ? (w1hi >> (sh - 32)) : __builtin_amdgcn_alignbit(w1hi, w1lo, sh)) & amask;
                LDSP(double) r1 = Tt + (size_t)(j * A + (int)a1) * RS + sb;
#pragma unroll
                for (int x = 0; x < 4; x++) p1[x] *= r1[x];
                if (any2) {
                  const uint32_t a2 = (sh >= 32 ? (w2hi >> (sh - 32)) : __builtin_amdgcn_alignbit(w2hi, w2lo, sh)) & amask;
                  LDSP(double) r2 = Tt + (size_t)(j * A + (int)a2) * RS + sb;
#pragma unroll
                  for (int x = 0; x < 4; x++) p2[x] *= r2[x];
                }
              }
              double acc[4];
#pragma unroll
              for (int x = 0; x < 4; x++) acc[x] = 0.0;
#pragma unroll
              for (int h = 0; h < KT; h++) {
#pragma unroll
                for (int x = 0; x < 4; x++) {
                  const double b = bp[h * slots + sb + x];
                  const double ph = (h == h1) ? p1[x] : ((h == h2) ? p2[x] : b);
                  acc[x] += ph * invK;
                }
              }
              double blk = 0.0;
#pragma unroll
              for (int x = 0; x < 4; x++)
                if (x < nb) blk += read_log(acc[x]) * cwt[sb + x];
              s_l += blk;
            }
            // the leaf of lane l joins the butterfly's tree: a binary counter of partial sums
            double v = s_l;
            int lvl = 0;
            for (int pp = p; pp & 1; pp >>= 1, lvl++) v = stk[lvl * WAVE + lane] + v;
            stk[lvl * WAVE + lane] = v;
          }
        }
        if (valid) ullk[uid] = stk[6 * WAVE + lane];
      }
      lds_sync();

      // ---- (4) option probabilities, then the totals a visit without a move would have formed ----
      for (int base = 0; base < n_slots; base += WAVE) {
        const int s = base + lane;
        if (s < n_slots) {
          GWords<KT> pw;
          uint32_t oin, lo;
          int my_no;
          slot_request(s, pw, oin, lo, my_no);
          const double llk_i = ullk[sluid[s]];
          double lprior_ratio = 0.0;
          if (!isnan(inbreeding)) lprior_ratio = prior_of<KT>(pt, inbreeding, dosage_of_labels(oin, lo, KT, true)) - lprior_cur;
          const int n_return = step_type == 0 ? recombination_n_options(oin, lo, KT) : dosage_n_options(oin, lo, KT);
          const double lproposal_ratio = lninv[n_return] - lninv[my_no];
          const double mh = ((llk_i - cur_llk) + lprior_ratio) * temp + lproposal_ratio;
          ptab[s] = exp(fmin(0.0, mh) - ln[my_no]);
        }
      }
      lds_sync();
      if (lane < n_iv) {
        const int no2 = ivno[lane], off2 = ivoff[lane];
        if (no2 > 0) {
          double cacc = 0.0;
          for (int o = 0; o < no2; o++) cacc += ptab[off2 + o];
          mtot[ivse[lane] >> 16] = cacc;
        }
      }
      lds_sync();
    }
  }
}

}  // namespace mchap
