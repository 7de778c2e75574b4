// Table completion of the phased de-novo sampler (kernel 5) for MI355X (gfx950): ONE WORKGROUP PER CHAIN, ONE WAVEFRONT PER
// REQUEST, nothing but the completion in the kernel.
//
// Before a settled chain can coast (denovo_coast_kernel.hpp) the total move probability of every interval step
// (step type, start, stop) of its current genotype must be known (structural.py:433-673: the options of an interval, their
// Metropolis-Hastings ratios, the sum of their probabilities).  Until round 4 the exporting launch of denovo_spec_kernel<.., true>
// completed that table itself, with the code of a visit: 64 option slots per round, every distinct request of a round evaluated
// one after the other by the chain's only wavefront -- at 256 VGPRs, two wavefronts per SIMD.  That was 3.0 ms of the 12.5 ms of
// a sampler call at BASELINE configs[1] (exporting launch 8.97 ms with the completion, 5.93 ms without).
//
// Here the completion is its own launch (the exporting launch runs with PIPE_NOFILL):
//   0. set-up, once per chain: the unit's table for the first four read chunks as float64 in LDS ([position x allele][read],
//      decoded from the coded table and the unit's dictionary; 32 KB at configs[1]; when it does not fit, the evaluations read
//      the coded table as denovo_spec_kernel does), the read weights, and the haplotype products of the chain's current
//      genotype (already divided by the ploidy) -- one haplotype per wavefront;
//   1. listing: one THREAD per (step type, interval) entry -- labels inside / outside the interval, option count; entries without
//      options are settled at once (-1: the step consumes no draw).  A block-wide prefix sum of the option counts cuts the
//      entries into chunks of at most FILLW_SLOTS option slots (BASELINE configs[1]: all 648 slots of a chain in one chunk);
//   2. one thread per option slot forms its proposal genotype (the reference's enumeration order, the code of denovo_spec_kernel)
//      and the slots are de-duplicated through an LDS hash table by atomicCAS: options of different intervals coincide (~650
//      slots are ~170 distinct genotypes).  Which slot represents a genotype is a race; the value does not depend on it.  One
//      thread per distinct request then probes the chain's likelihood cache (what the chain's own steps evaluated: a hit is the
//      value this very arithmetic produced, bit for bit);
//   3. the workgroup's wavefronts take the remaining requests round-robin, ONE WAVEFRONT PER REQUEST, lanes over reads: the
//      products of the one or two haplotype words the request changed -- factors in position order from 1.0 --, the haplotype
//      terms added in haplotype order, read_log, the read weights, wave_sum: the same factors, products, sums and butterfly as
//      spec_coop_reuse, so the value is that of the in-kernel completion bit for bit;
//   4. one thread per slot forms its option's probability, one thread per interval adds them in option order: the totals a visit
//      without a move would have formed.
// The kernel holds no stepping code: it compiles to <= 128 VGPRs (four wavefronts per SIMD; profiles/r04_kernel_resource_usage.txt).
// Results: tests/test_gpu_fillw.py compares the tables and the traces with the in-kernel completion (tuning flag 1024 switches this
// kernel off; 2048: no cache probe; 4096: no table in LDS -- same tables in every combination).
//
// Launch: grid = chains of the list (PipeState records written by the exporting launch), block = 64 * FILLW_NW threads, dynamic
// LDS fillw_lds_bytes().  Shapes: packed genotype of at most 64 bits (K x bits x positions; else the in-kernel completion runs).
#pragma once
#include "denovo_spec_kernel.hpp"

namespace mchap {

#ifndef MCHAP_FILLW_NW
#define MCHAP_FILLW_NW 8   // wavefronts per workgroup (= per chain)
#endif
#ifndef MCHAP_FILLW_WPE
#define MCHAP_FILLW_WPE 4  // wavefronts per SIMD the kernel is compiled for (128 VGPRs)
#endif
constexpr int FILLW_NW = MCHAP_FILLW_NW;
constexpr int FILLW_NT = 64 * FILLW_NW;
constexpr int FILLW_ENT = 256;     // entries listed per pass (one thread each: an entry of a chunk is named by one byte)
constexpr int FILLW_SLOTS = 768;   // option slots of one chunk of intervals (packed keys: configs[1]'s 648 slots are one chunk)
constexpr int FILLW_SLOTS_WIDE = 512;  // ... for genotypes wider than 64 bits: a slot's key is its one or two changed words
constexpr int FILLW_HASH = 2048;   // open-addressing slots of a chunk's request table (uint32: slot + 1 of the representative)
constexpr int FILLW_TAB_BYTES = 32 * 1024;  // the unit's float64 table in LDS, when it fits
static_assert(FILLW_NT >= FILLW_ENT, "one thread per listed entry");

// rows of the LDS table (0: it does not fit; the evaluations then read the coded table) for a batch's dimensions
__host__ __device__ inline int fillw_tab_rows(int max_pos, int max_allele, int rpad) {
  const int nb = rpad / 64 < 4 ? rpad / 64 : 4;
  const int rows = max_pos * max_allele;
  return (size_t)rows * nb * 64 * 8 <= (size_t)FILLW_TAB_BYTES ? rows : 0;
}

__host__ __device__ inline int fillw_slots(bool wide) { return wide ? FILLW_SLOTS_WIDE : FILLW_SLOTS; }
struct FillwLds {
  size_t mk1, mk2, mllk, mhh, mtab;  // (memo of evaluated requests across a chain's chunks: memo_cap > 0 only)
  size_t skey2, shh;  // (wide keys only)
  size_t tab, bp, cw, dict, pt, ln, lninv, skey, sllk, htab, urep, umiss, elin, elout, ese, eoff, eno, slrep, slent, slopt, cols, shift, scal, total;
};
// memo_cap: entries of the memo of evaluated requests kept ACROSS a chain's chunks (0: none): the in-kernel completion found a
// genotype that recurs in a later round in the chain's likelihood cache; this kernel's chunks are de-duplicated one by one, so at
// big shapes (configs[4]: 17 000 slots in 34 chunks, 3 200 distinct genotypes) the same request used to be evaluated in several
// chunks (its first evaluation may not even stay in the cache: the blind way a store picks).  The memo is exact (full keys).
__host__ __device__ inline int fillw_memo_cap(int K, int max_pos, int tab_rows, bool wide) {
  if (K >= 5) return 2048;                      // (one workgroup per CU at these ploidies anyway: the LDS is there)
  return (wide && tab_rows == 0) ? 1024 : 0;    // two workgroups per CU: only beside a layout without the float64 table
}
__host__ __device__ inline FillwLds fillw_lds(int K, int max_pos, int tab_rows, int rpad, bool wide = false, int memo_cap = 0) {
  FillwLds L;
  const int nb = rpad / 64 < 4 ? rpad / 64 : 4;
  const size_t NS = (size_t)fillw_slots(wide);
  size_t o = 0;
  L.tab = o; o += (size_t)8 * tab_rows * nb * 64;  // [position x allele][chunk][lane] float64 factors, or nothing
  L.bp = o; o += (size_t)8 * K * 4 * 64;          // haplotype products / K of the current genotype, first four read chunks
  L.cw = o; o += (size_t)8 * 4 * 64;              // read weights of those chunks
  L.dict = o; o += (size_t)8 * DICT_MAX;
  L.pt = o; o += (size_t)8 * (2 * K + 5);
  L.ln = o; o += (size_t)8 * SPEC_LN;
  L.lninv = o; o += (size_t)8 * SPEC_LN;
  L.skey = o; o += (size_t)8 * NS;       // packed proposal genotype of a slot; later the slot's probability
  L.skey2 = o; o += wide ? (size_t)8 * NS : 0;  // second changed word of a wide slot
  L.sllk = o; o += (size_t)8 * NS;       // log likelihood, at the representative's slot
  L.htab = o; o += (size_t)4 * FILLW_HASH;
  L.elin = o; o += (size_t)4 * FILLW_ENT;         // per entry of the chunk (= per listing thread): labels, interval, offset, options
  L.elout = o; o += (size_t)4 * FILLW_ENT;
  L.ese = o; o += (size_t)4 * FILLW_ENT;
  L.scal = o; o += (size_t)4 * 16;                // block scalars: wave totals of the scan, cut, slots, distinct requests, misses
  L.urep = o; o += (size_t)2 * NS;       // distinct request -> its representative slot
  L.umiss = o; o += (size_t)2 * NS;      // ... those the chain's cache does not hold
  L.slrep = o; o += (size_t)2 * NS;      // slot -> representative slot
  L.eoff = o; o += (size_t)2 * FILLW_ENT;
  L.eno = o; o += (size_t)2 * FILLW_ENT;
  L.cols = o; o += (size_t)2 * max_pos;
  L.slent = o; o += NS;
  L.slopt = o; o += NS;
  L.shh = o; o += wide ? NS : 0;  // (h1 << 4) | h2 of a wide slot's changed haplotypes (h2 = 15: one only)
  L.shift = o; o += (size_t)max_pos;
  o = (o + 7) & ~(size_t)7;
  L.mk1 = o; o += (size_t)8 * memo_cap;
  L.mk2 = o; o += (size_t)8 * memo_cap;
  L.mllk = o; o += (size_t)8 * memo_cap;
  L.mtab = o; o += (size_t)4 * 2 * memo_cap;
  L.mhh = o; o += (size_t)memo_cap;
  L.total = (o + 63) & ~(size_t)63;
  return L;
}
__host__ __device__ inline size_t fillw_lds_bytes(int K, int max_pos, int tab_rows, int rpad, bool wide = false, int memo_cap = 0) {
  return fillw_lds(K, max_pos, tab_rows, rpad, wide, memo_cap).total;
}
// Shapes the kernel takes: ploidy 2..8, a haplotype word of at most 64 bits (every shape the phased sampler runs).  The packed
// genotype is a request's key while it fits 64 bits; beyond that (fillw_wide) a request is keyed by the one or two haplotype words
// it changes (the genotype it proposes differs from the chain's in those only).
__host__ __device__ inline bool fillw_takes(int K, int max_pos, int max_allele) {
  return K >= 2 && K <= 8 && allele_bits(max_allele) * max_pos <= 64 && max_pos <= 64;
}
__host__ __device__ inline bool fillw_wide(int K, int max_pos, int max_allele) { return K * allele_bits(max_allele) * max_pos > 64; }

// (start, stop) of entry e of an interval table: e = stop (stop - 1) / 2 + start, 0 <= start < stop
__device__ __forceinline__ void fillw_entry_interval(int e, int &start, int &stop) {
  int s = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);
  while (s * (s - 1) / 2 > e) s--;
  while ((s + 1) * s / 2 <= e) s++;
  stop = s;
  start = e - s * (s - 1) / 2;
}

// The my_o-th option of interval labels (lin, lout) in the reference's enumeration order (structural.py:121-178 /
// 240-307; the code of denovo_spec_kernel's spec_structural): the `in` label pack after the move.
template <int KT>
__device__ __forceinline__ uint32_t fillw_option(int step_type, uint32_t lin, uint32_t lout, int my_o) {
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

// Rows of the table for the Mh positions of ONE haplotype word (wave-uniform): lane j holds the row of position j, which is what
// spec_hap_prod reads through v_readlane (pairs p0 = 0 .. Mh - 1 of a PairRows).  Mh <= 64.
__device__ __forceinline__ PairRows fillw_rows(uint64_t w, LDSP(const uint16_t) cols, LDSP(const uint8_t) shift, int Mh, uint32_t amask, int lane) {
  PairRows R;
  R.r0 = 0;
  R.r1 = 0;
  R.r2 = 0;
  if (lane < Mh) R.r0 = (int)cols[lane] + (int)((uint32_t)(w >> shift[lane]) & amask);
  return R;
}

// The same product from the float64 rows (units without a coded table: more than DICT_MAX distinct values), factors in position
// order from 1.0 as spec_coop_body forms them
template <int RPL>
__device__ __forceinline__ void fillw_hap_prod_rows(const PairRows rows, int Mh, GLBP(const double) rt, int rpad, double (&prod)[RPL]) {
  constexpr int UNR = 4;
#pragma unroll
  for (int i = 0; i < RPL; i++) prod[i] = 1.0;
  for (int j0 = 0; j0 < Mh; j0 += UNR) {
    double v[UNR][RPL];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const int row = __builtin_amdgcn_readlane(rows.r0, min(j0 + u, Mh - 1));
      GLBP(const double) rp = rt + (size_t)row * rpad;
#pragma unroll
      for (int i = 0; i < RPL; i++) v[u][i] = rp[WAVE * i];
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const bool on = j0 + u < Mh;
#pragma unroll
      for (int i = 0; i < RPL; i++) prod[i] *= on ? v[u][i] : 1.0;
    }
  }
}

struct FillwUnit {
  LDSP(double) dict;
  LDSP(const uint16_t) cols;
  LDSP(const uint8_t) shift;
  GLBP(const uint8_t) ct;   // the lane's first code (coded table)
  GLBP(const double) rt;    // the lane's first read of the float64 rows
  GLBP(const double) cw;    // the lane's first read weight
  LDSP(const double) tab;   // the table of the first block in LDS ([sampled position x allele][chunk][lane]) + lane, or null
  LDSP(const double) bpk;   // haplotype products / K of the current genotype, first block ([h][4][64]) + lane, or null
  LDSP(const double) cwl;   // read weights of the first block + lane
  GLBP(double) gbp;         // deep units: the chain's [K][rpad] rows of haplotype products for the chunks beyond the fourth + lane, or null
  int Mh, A, bits, crow, rpad, nch, tab_rs;
  uint32_t amask;
  bool coded;
  bool has_gbp;
  bool has_tab, has_bpk;  // (explicit flags: an LDS pointer at offset 0 -- lane 0's `tab + lane` -- must not read as "no table")
  bool w01;               // the unit's read weights are 0 / 1: one logarithm per lane and block of chunks (read_log_sum)
};

// prod[i] = product over the positions of haplotype word w, reads lane + 64 (cb + i), from the coded table (or the float64 rows)
template <int RPL, class CT>
__device__ __forceinline__ void fillw_hap(const FillwUnit &U, uint64_t w, int cb, int lane, double (&prod)[RPL]) {
  const PairRows rows = fillw_rows(w, U.cols, U.shift, U.Mh, U.amask, lane);
  if (U.coded) spec_hap_prod<RPL, CT, false>(U.dict, rows, 0, U.Mh, U.ct + cb, U.crow, prod);
  else fillw_hap_prod_rows<RPL>(rows, U.Mh, U.rt + (size_t)cb * WAVE, U.rpad, prod);
}
// ... of the first block, from the table in LDS: the same factors (dict[code], or the float64 row's entry) in the same order
template <int RPL>
__device__ __forceinline__ void fillw_hap_tab(const FillwUnit &U, uint64_t w, double (&prod)[RPL]) {
  const uint32_t wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w);
  const uint32_t whi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
  const uint64_t ws = ((uint64_t)whi << 32) | wlo;  // (wave-uniform: the allele extraction runs on the scalar unit)
#pragma unroll
  for (int i = 0; i < RPL; i++) prod[i] = 1.0;
  int sh = U.bits * (U.Mh - 1);
#pragma unroll 4
  for (int j = 0; j < U.Mh; j++, sh -= U.bits) {
    const int row = j * U.A + (int)((uint32_t)(ws >> sh) & U.amask);
    LDSP(const double) rp = U.tab + row * U.tab_rs;
#pragma unroll
    for (int i = 0; i < RPL; i++) prod[i] *= rp[WAVE * i];
  }
}

// One block of RPL read chunks of a request that differs from the current genotype g in haplotypes h1 (word w1) and, when
// h2 >= 0, h2 (word w2): the lane's partial sum  sum_i read_log(sum_h prod_h / K) * weight  -- haplotype terms added in haplotype
// order, chunks in chunk order, as spec_coop_reuse / spec_coop_coded / spec_coop_body do.  FIRST: the block of chunks 0..RPL-1,
// whose base terms (and, when it fits, table) live in LDS; else every haplotype's product is formed from the coded table.
template <int KT, int RPL, class CT, bool FIRST, bool DEEP>
__device__ __forceinline__ double fillw_block(const FillwUnit &U, const GWords<KT> g, int h1, uint64_t w1, int h2, uint64_t w2, int cb,
                                              int lane) {
  const double invK = 1.0 / (double)KT;
  double acc[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) acc[i] = 0.0;
  if (FIRST && U.has_bpk) {
    double p1[RPL], p2[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i++) p2[i] = 1.0;
    if (U.has_tab) {
      fillw_hap_tab<RPL>(U, w1, p1);
      if (h2 >= 0) fillw_hap_tab<RPL>(U, w2, p2);
    } else {
      fillw_hap<RPL, CT>(U, w1, 0, lane, p1);
      if (h2 >= 0) fillw_hap<RPL, CT>(U, w2, 0, lane, p2);
    }
#pragma unroll
    for (int i = 0; i < RPL; i++) {
      p1[i] *= invK;
      p2[i] *= invK;
    }
#pragma unroll
    for (int h = 0; h < KT; h++) {
#pragma unroll
      for (int i = 0; i < RPL; i++) {
        const double b = U.bpk[(h * 4 + i) * WAVE];
        acc[i] += (h == h1) ? p1[i] : ((h == h2) ? p2[i] : b);
      }
    }
    double wv[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i++) wv[i] = U.cwl[i * WAVE];
    return read_log_sum<RPL>(acc, wv, U.w01);
  }
  if (DEEP && !FIRST && U.has_gbp) {
    // deep units: the products of the haplotypes the request did not change come from the chain's rows in the workspace (formed
    // once per chain in the set-up: spec_coop_reuse_g's arithmetic) -- all K x RPL loads go out together, ahead of the products
    double bpv[KT][RPL];
#pragma unroll
    for (int h = 0; h < KT; h++)
#pragma unroll
      for (int i = 0; i < RPL; i++) bpv[h][i] = U.gbp[(size_t)h * U.rpad + (size_t)(cb + i) * WAVE];
    double p1[RPL], p2[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i++) p2[i] = 1.0;
    fillw_hap<RPL, CT>(U, w1, cb, lane, p1);
    if (h2 >= 0) fillw_hap<RPL, CT>(U, w2, cb, lane, p2);
#pragma unroll
    for (int h = 0; h < KT; h++)
#pragma unroll
      for (int i = 0; i < RPL; i++) acc[i] += ((h == h1) ? p1[i] : ((h == h2) ? p2[i] : bpv[h][i])) * invK;
    double wv[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i++) wv[i] = U.cw[(size_t)(cb + i) * WAVE];
    return read_log_sum<RPL>(acc, wv, U.w01);
  }
#pragma unroll 1
  for (int h = 0; h < KT; h++) {
    const uint64_t w = (h == h1) ? w1 : ((h == h2) ? w2 : sel_word<KT>(g, h));
    double ph[RPL];
    fillw_hap<RPL, CT>(U, w, cb, lane, ph);
#pragma unroll
    for (int i = 0; i < RPL; i++) acc[i] += ph[i] * invK;
  }
  double wv[RPL];
#pragma unroll
  for (int i = 0; i < RPL; i++) wv[i] = U.cw[(size_t)(cb + i) * WAVE];
  return read_log_sum<RPL>(acc, wv, U.w01);
}

template <int KT, bool FIRST, bool DEEP>
__device__ __forceinline__ double fillw_blocks(const FillwUnit &U, const GWords<KT> g, int h1, uint64_t w1, int h2, uint64_t w2, int cb,
                                               int nb, int lane) {
  if (nb >= 4) return fillw_block<KT, 4, uint32_t, FIRST, DEEP>(U, g, h1, w1, h2, w2, cb, lane);
  if (nb == 3) return fillw_block<KT, 3, uint32_t, FIRST, DEEP>(U, g, h1, w1, h2, w2, cb, lane);
  if (nb == 2) return fillw_block<KT, 2, uint16_t, FIRST, DEEP>(U, g, h1, w1, h2, w2, cb, lane);
  return fillw_block<KT, 1, uint8_t, FIRST, DEEP>(U, g, h1, w1, h2, w2, cb, lane);
}

// WIDE: genotypes of more than 64 bits (requests keyed by their changed words); DEEP: units of more than four read chunks keep
// the current genotype's products of the later chunks in the chain's workspace rows.  Separate instantiations: the paths of one
// must not cost the other its registers (BASELINE configs[1] runs <K, false, false>).
template <int KT, bool WIDE = false, bool DEEP = false>
__global__ __launch_bounds__(FILLW_NT, MCHAP_FILLW_WPE) void denovo_fillw_kernel(const SimtParams P) {
  constexpr int NSLOTS = WIDE ? FILLW_SLOTS_WIDE : FILLW_SLOTS;
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);  // (wave-uniform: the loops over requests / haplotypes are scalar)
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
  const mchap_unit Un = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const double *mf = P.meta_f + (size_t)u * meta_f_stride(P.max_ploidy, P.max_pos, P.max_allele);
  if (mi[META_I_STATUS] != MCHAP_UNIT_OK) return;
  const int Mh = mi[META_I_MH];
  const int A = Un.max_allele;
  const int bits = allele_bits(A);
  const uint32_t amask = (1u << bits) - 1u;
  const int rpad = D.rpad;
  const int key_bits = bits * Mh;
  const uint64_t key_mask = key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull);
  const double inbreeding = Un.inbreeding;
  const double temp = D.temps[0];
  const bool flat = mi[META_I_FLAT] != 0 && !(P.flags & 128);
  const int tab_rows = (P.flags & 4096) ? 0 : P.fill_lt;  // rows the LDS table was sized for (0: none)
  const int nbmax = rpad / WAVE < 4 ? rpad / WAVE : 4;
  const int memo_cap = (P.fill_kw >> 8) * 256;  // entries of the cross-chunk memo (host: fillw_memo_cap)
  const FillwLds L = fillw_lds(KT, P.max_pos, P.fill_lt, rpad, WIDE, memo_cap);
  LDSP(double) tab = lds_cast<double>(smem + L.tab);
  LDSP(double) bp = lds_cast<double>(smem + L.bp);
  LDSP(double) cwl = lds_cast<double>(smem + L.cw);
  LDSP(double) dict = lds_cast<double>(smem + L.dict);
  LDSP(double) pt = lds_cast<double>(smem + L.pt);
  LDSP(double) ln = lds_cast<double>(smem + L.ln);
  LDSP(double) lninv = lds_cast<double>(smem + L.lninv);
  LDSP(uint64_t) skey = lds_cast<uint64_t>(smem + L.skey);     // packed genotype, or (WIDE) the first changed word
  LDSP(uint64_t) skey2 = lds_cast<uint64_t>(smem + L.skey2);   // (WIDE) the second changed word
  LDSP(uint8_t) shh = lds_cast<uint8_t>(smem + L.shh);         // (WIDE) which haplotypes: (h1 << 4) | h2, h2 = 15 when one only
  LDSP(uint64_t) mk1 = lds_cast<uint64_t>(smem + L.mk1);       // memo across chunks: key, second word, haplotypes, llk
  LDSP(uint64_t) mk2 = lds_cast<uint64_t>(smem + L.mk2);
  LDSP(double) mllk = lds_cast<double>(smem + L.mllk);
  LDSP(uint8_t) mhh = lds_cast<uint8_t>(smem + L.mhh);
  unsigned int *mtab = reinterpret_cast<unsigned int *>(smem + L.mtab);  // [2 memo_cap] open addressing: entry + 1
  for (int i = tid; i < 2 * memo_cap; i += FILLW_NT) mtab[i] = 0u;
  if (tid == 0) reinterpret_cast<int *>(smem + L.scal)[12] = 0;
  LDSP(double) ptab = lds_cast<double>(smem + L.skey);  // (after the evaluation: the keys are no longer needed)
  LDSP(double) sllk = lds_cast<double>(smem + L.sllk);
  unsigned int *htab = reinterpret_cast<unsigned int *>(smem + L.htab);
  LDSP(uint32_t) elin = lds_cast<uint32_t>(smem + L.elin);
  LDSP(uint32_t) elout = lds_cast<uint32_t>(smem + L.elout);
  LDSP(uint32_t) ese = lds_cast<uint32_t>(smem + L.ese);
  int *scal = reinterpret_cast<int *>(smem + L.scal);  // [0..4) wave totals, [8] cut, [9] slots, [10] distinct requests, [11] misses
  LDSP(uint16_t) urep = lds_cast<uint16_t>(smem + L.urep);
  LDSP(uint16_t) umiss = lds_cast<uint16_t>(smem + L.umiss);
  LDSP(uint16_t) slrep = lds_cast<uint16_t>(smem + L.slrep);
  LDSP(uint16_t) eoff = lds_cast<uint16_t>(smem + L.eoff);
  LDSP(uint16_t) eno = lds_cast<uint16_t>(smem + L.eno);
  LDSP(uint16_t) cols = lds_cast<uint16_t>(smem + L.cols);
  LDSP(uint8_t) slent = lds_cast<uint8_t>(smem + L.slent);
  LDSP(uint8_t) slopt = lds_cast<uint8_t>(smem + L.slopt);
  LDSP(uint8_t) shift = lds_cast<uint8_t>(smem + L.shift);

  for (int i = tid; i < SPEC_LN; i += FILLW_NT) {
    ln[i] = c_ln[i];
    lninv[i] = c_ln_inv[i];
  }
  for (int j = tid; j < Mh; j += FILLW_NT) {
    cols[j] = (uint16_t)mi[META_I_COLS + j];
    shift[j] = (uint8_t)(bits * (Mh - 1 - j));
  }
  if (!isnan(inbreeding))
    for (int i = tid; i < 2 * KT + 5; i += FILLW_NT) pt[i] = mf[meta_f_prior(0) + i];
  const int nd = (P.flags & 4) ? 0 : mi[META_I_NDICT];
  {
    const double *du = P.dict + (size_t)u * DICT_MAX;
    for (int i = tid; i < nd; i += FILLW_NT) dict[i] = du[i];
  }
  GWords<KT> g;  // the chain's current genotype, in the chain's own haplotype order
#pragma unroll
  for (int h = 0; h < KT; h++) g.w[h] = st->g[h];
  const double cur_llk = st->llk;
  double *memo = P.pipe_memo + (size_t)q * 2 * E;
  const uint64_t full = mask_of(bits, Mh, 0, Mh);
  const int n_entries = spec_memo_entries(Mh);
  FillwUnit FU;
  FU.dict = dict;
  FU.cols = cols;
  FU.shift = shift;
  FU.coded = nd != 0;
  FU.ct = (GLBP(const uint8_t))(P.codes + (size_t)u * P.max_ma * WAVE * P.cstride) + (size_t)lane * P.cstride;
  FU.rt = (GLBP(const double))(P.rt + (size_t)u * P.max_ma * rpad) + lane;
  FU.cw = (GLBP(const double))(P.cntw + (size_t)u * rpad) + lane;
  FU.Mh = Mh;
  FU.A = A;
  FU.bits = bits;
  FU.crow = WAVE * P.cstride;
  FU.rpad = rpad;
  // read chunks of THIS unit (a batch is padded to its deepest unit; the chunks beyond a unit's own reads hold padding only --
  // weight 0, terms +-0.0 --: as spec_coop_all, they are not evaluated)
  FU.nch = max(1, min(rpad / WAVE, ((int)Un.n_reads + WAVE - 1) / WAVE));
  FU.amask = amask;
  FU.tab_rs = nbmax * WAVE;
  const int nb0 = FU.nch < 4 ? FU.nch : 4;
  const bool use_tab = !flat && Mh * A <= tab_rows;
  FU.tab = (LDSP(const double))(tab + lane);
  FU.bpk = (LDSP(const double))(bp + lane);
  FU.w01 = mi[META_I_W01] != 0;
  FU.has_tab = use_tab;
  FU.has_bpk = !flat;
  FU.cwl = (LDSP(const double))(cwl + lane);
  __syncthreads();
  const double lprior_cur = isnan(inbreeding) ? 0.0 : prior_of<KT>(pt, inbreeding, dosage_words<KT>(g));
  if (!flat) {
    // the first block's read weights, and the table of the sampled positions: entry (j, a, read) = dict[code] (or the float64
    // row's entry when the unit has no coded table): the very doubles the evaluations of denovo_spec_kernel gather
    for (int i = tid; i < nb0 * WAVE; i += FILLW_NT) cwl[i] = (P.cntw + (size_t)u * rpad)[i];
    if (use_tab) {
      const int per_row = nb0 * WAVE;
      const int n_el = Mh * A * per_row;
      const uint8_t *cu = P.codes + (size_t)u * P.max_ma * WAVE * P.cstride;
      const double *ru = P.rt + (size_t)u * P.max_ma * rpad;
      for (int x = tid; x < n_el; x += FILLW_NT) {
        const int row = x / per_row, rem = x - row * per_row;
        const int i = rem / WAVE, l = rem - i * WAVE;
        const int j = row / A, a = row - j * A;
        const int grow = (int)cols[j] + a;
        double v;
        if (nd != 0) v = dict[cu[((size_t)grow * WAVE + l) * P.cstride + i]];
        else v = ru[(size_t)grow * rpad + l + WAVE * i];
        tab[row * FU.tab_rs + i * WAVE + l] = v;
      }
      __syncthreads();
    }
    // the current genotype's haplotype products of the first block, divided by the ploidy (the term a likelihood adds per
    // haplotype): wavefront w forms haplotypes w, w + NW, ..
    const double invK = 1.0 / (double)KT;
    for (int h = wv; h < KT; h += FILLW_NW) {
      const uint64_t w = sel_word<KT>(g, h);
      double ph[4] = {1.0, 1.0, 1.0, 1.0};
      if (use_tab) {
        if (nb0 == 4) {
          fillw_hap_tab<4>(FU, w, ph);
        } else if (nb0 == 3) {
          double p3[3];
          fillw_hap_tab<3>(FU, w, p3);
          ph[0] = p3[0]; ph[1] = p3[1]; ph[2] = p3[2];
        } else if (nb0 == 2) {
          double p2[2];
          fillw_hap_tab<2>(FU, w, p2);
          ph[0] = p2[0]; ph[1] = p2[1];
        } else {
          double p1[1];
          fillw_hap_tab<1>(FU, w, p1);
          ph[0] = p1[0];
        }
      } else if (nb0 == 4) {
        fillw_hap<4, uint32_t>(FU, w, 0, lane, ph);
      } else if (nb0 == 3) {
        double p3[3];
        fillw_hap<3, uint32_t>(FU, w, 0, lane, p3);
        ph[0] = p3[0]; ph[1] = p3[1]; ph[2] = p3[2];
      } else if (nb0 == 2) {
        double p2[2];
        fillw_hap<2, uint16_t>(FU, w, 0, lane, p2);
        ph[0] = p2[0]; ph[1] = p2[1];
      } else {
        double p1[1];
        fillw_hap<1, uint8_t>(FU, w, 0, lane, p1);
        ph[0] = p1[0];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) bp[(h * 4 + i) * WAVE + lane] = ph[i] * invK;
    }
  }
  // deep units (more than four read chunks): the current genotype's haplotype products of the chunks beyond the fourth go to the
  // chain's rows in the workspace (SimtParams::gbp, as the deep instantiation of denovo_spec_kernel keeps them) -- formed once
  // here, so that a request forms only the products of the words it changed in every block of chunks
  FU.gbp = nullptr;
  FU.has_gbp = false;
  if (DEEP && !flat && FU.nch > 4 && P.gbp != nullptr && !(P.flags & 4096)) {
    GLBP(double) rows = (GLBP(double))(P.gbp + (size_t)q * P.max_ploidy * rpad) + lane;
    for (int h = wv; h < KT; h += FILLW_NW) {
      const uint64_t w = sel_word<KT>(g, h);
      for (int cb = 4; cb < FU.nch; cb += 4) {
        const int rem = FU.nch - cb;
        double ph[4] = {1.0, 1.0, 1.0, 1.0};
        if (rem >= 4) {
          fillw_hap<4, uint32_t>(FU, w, cb, lane, ph);
        } else if (rem == 3) {
          double p3[3];
          fillw_hap<3, uint32_t>(FU, w, cb, lane, p3);
          ph[0] = p3[0]; ph[1] = p3[1]; ph[2] = p3[2];
        } else if (rem == 2) {
          double p2[2];
          fillw_hap<2, uint16_t>(FU, w, cb, lane, p2);
          ph[0] = p2[0]; ph[1] = p2[1];
        } else {
          double p1[1];
          fillw_hap<1, uint8_t>(FU, w, cb, lane, p1);
          ph[0] = p1[0];
        }
        for (int i = 0; i < (rem < 4 ? rem : 4); i++) rows[(size_t)h * rpad + (size_t)(cb + i) * WAVE] = ph[i];
      }
    }
    FU.gbp = rows;
    FU.has_gbp = true;
    __syncthreads();  // (workgroup-scope release / acquire: the rows are read by every wavefront of the chain)
  }
  // the chain's likelihood cache (what its own steps evaluated), probed when the packed genotype is its tag (tag_of)
  // ... exact tags (the packed genotype) up to 63 bits; wider genotypes are tagged by a hash and verified against their words
  const uint64_t *ckeys = D.cache_keys ? D.cache_keys + (size_t)q * (size_t)D.cache_slots * D.cache_key_words : nullptr;
  const bool probe = D.cache_slots > 0 && !(P.flags & 2048) && !flat && (WIDE ? (ckeys != nullptr && KT * key_bits > 63) : KT * key_bits <= 63);
  const ulonglong2 *cache = reinterpret_cast<const ulonglong2 *>(D.cache) + (size_t)q * (size_t)D.cache_slots;
  const uint32_t cache_mask = probe ? (uint32_t)(D.cache_slots / 8) - 1u : 0u;

  int cursor = 0;  // (step type, entry) pairs are walked as one sequence ge = step_type * n_entries + e
  while (cursor < 2 * n_entries) {
    // ---- (1) list: one thread per entry; prefix sum of the option counts; the chunk is the longest prefix that fits ----
    for (int i = tid; i < FILLW_HASH; i += FILLW_NT) htab[i] = 0u;
    if (tid == 0) {
      scal[8] = FILLW_ENT;
      scal[9] = 0;
      scal[10] = 0;
      scal[11] = 0;
    }
    const int ge = cursor + tid;
    const int ty = ge >= n_entries ? 1 : 0;
    const int e = ge - ty * n_entries;
    double *mt = memo + (size_t)ty * E;
    bool unk = tid < FILLW_ENT && ge < 2 * n_entries && isnan(mt[e]);
    uint32_t lin = 0, lout = 0;
    int no = 0, start = 0, stop = 1;
    if (unk) {
      fillw_entry_interval(e, start, stop);
      const uint64_t min_ = mask_of(bits, Mh, start, stop);
      lin = seg_labels<KT>(g, min_);
      lout = seg_labels<KT>(g, full & ~min_);
      no = ty == 0 ? recombination_n_options(lin, lout, KT) : dosage_n_options(lin, lout, KT);
    }
    int incl = no;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const int v = __shfl_up(incl, o, WAVE);
      if (lane >= o) incl += v;
    }
    if (lane == WAVE - 1 && wv < FILLW_ENT / WAVE) scal[wv] = incl;
    __syncthreads();
    int excl = incl - no;
    for (int w = 0; w < wv && w < FILLW_ENT / WAVE; w++) excl += scal[w];
    if (unk && no > 0 && excl + no > NSLOTS) atomicMin(&scal[8], tid);
    __syncthreads();
    const int cut = scal[8];
    const bool mine = unk && tid < cut;
    if (mine && no == 0) mt[e] = -1.0;  // no options: the step consumes no draw
    const bool listed = mine && no > 0;
    if (listed) {
      elin[tid] = lin;
      elout[tid] = lout;
      ese[tid] = (uint32_t)start | ((uint32_t)stop << 8) | ((uint32_t)e << 16) | ((uint32_t)ty << 31);
      eoff[tid] = (uint16_t)excl;
      eno[tid] = (uint16_t)no;
      for (int o = 0; o < no; o++) {
        slent[excl + o] = (uint8_t)tid;
        slopt[excl + o] = (uint8_t)o;
      }
      atomicMax(&scal[9], excl + no);
    }
    __syncthreads();
    const int n_slots = scal[9];
    cursor += cut;
    if (n_slots == 0) continue;  // (block-uniform)

    // ---- (2) one thread per slot: the proposal genotype, packed; de-duplication through the hash table ----
    auto slot_option = [&](int s, uint32_t &oin, uint32_t &lo, int &my_no, int &ty_s, uint64_t &min_) {
      const int t = (int)slent[s];
      const uint32_t se = ese[t];
      ty_s = (int)(se >> 31);
      min_ = mask_of(bits, Mh, (int)(se & 255u), (int)((se >> 8) & 255u));
      lo = elout[t];
      my_no = eno[t];
      oin = fillw_option<KT>(ty_s, elin[t], lo, (int)slopt[s]);
    };
    for (int s = tid; s < n_slots; s += FILLW_NT) {
      uint32_t oin, lo;
      int my_no, ty_s;
      uint64_t min_;
      slot_option(s, oin, lo, my_no, ty_s, min_);
      if constexpr (!WIDE) {
        uint64_t key = 0;
#pragma unroll
        for (int h = 0; h < KT; h++) {
          const uint64_t wh = (g.w[h] & ~min_) | (sel_word<KT>(g, (int)nib(oin, h)) & min_);
          key = (key << key_bits) | wh;  // (K words of key_bits bits, together at most 64)
        }
        skey[s] = key;
      } else {
        // the one or two words the option changes, in haplotype order: the proposal is the chain's genotype but for them
        int h1 = 15, h2 = 15;
        uint64_t w1 = 0, w2 = 0;
#pragma unroll
        for (int h = 0; h < KT; h++) {
          const uint64_t wh = (g.w[h] & ~min_) | (sel_word<KT>(g, (int)nib(oin, h)) & min_);
          if (wh != g.w[h]) {
            if (h1 == 15) {
              h1 = h;
              w1 = wh;
            } else {
              h2 = h;
              w2 = wh;
            }
          }
        }
        if (h1 == 15) {  // (an option that changes nothing cannot occur; keyed as haplotype 0 unchanged)
          h1 = 0;
          w1 = g.w[0];
        }
        skey[s] = w1;
        skey2[s] = w2;
        shh[s] = (uint8_t)((h1 << 4) | h2);
      }
    }
    __syncthreads();
    for (int s = tid; s < n_slots; s += FILLW_NT) {
      const uint64_t key = skey[s];
      uint64_t key2 = 0;
      uint8_t hh = 0;
      uint64_t hk = key;
      if constexpr (WIDE) {
        key2 = skey2[s];
        hh = shh[s];
        hk = mix64(key ^ mix64(key2 + hh));
      }
      uint32_t hsh = (uint32_t)hk ^ ((uint32_t)(hk >> 32) * 0x9E3779B1u);
      hsh ^= hsh >> 16;
      hsh *= 0x7FEB352Du;
      hsh ^= hsh >> 15;
      int rep = -1;
      for (int pr = 0; pr < FILLW_HASH && rep < 0; pr++) {
        const int hi = (int)((hsh + (uint32_t)pr) & (uint32_t)(FILLW_HASH - 1));
        unsigned int v = htab[hi];
        if (v == 0u) v = atomicCAS(&htab[hi], 0u, (unsigned int)(s + 1));
        if (v == 0u) {
          rep = s;  // this slot represents its genotype
          urep[atomicAdd(&scal[10], 1)] = (uint16_t)s;
        } else if (skey[v - 1] == key && (!WIDE || (skey2[v - 1] == key2 && shh[v - 1] == hh))) {
          rep = (int)v - 1;
        }
      }
      slrep[s] = (uint16_t)rep;
    }
    __syncthreads();
    const int n_uniq = scal[10];
    // one thread per distinct request: the chain's cache (8-way sets, the hash of spec_eval); the misses are listed
    for (int uid = tid; uid < n_uniq; uid += FILLW_NT) {
      const int rep = (int)urep[uid];
      bool hit = false;
      if (flat) {
        sllk[rep] = cur_llk;  // a unit without information: every genotype has the chain's likelihood (spec_eval)
        hit = true;
      }
      if (!hit && memo_cap > 0) {  // evaluated for an earlier chunk of this chain
        const uint64_t k1 = skey[rep], k2 = WIDE ? skey2[rep] : 0ull;
        const uint8_t hh = WIDE ? shh[rep] : (uint8_t)0;
        const uint64_t hk = mix64(k1 ^ mix64(k2 + hh));
        const uint32_t m2 = (uint32_t)(2 * memo_cap - 1);
        for (uint32_t pr = 0, hi = (uint32_t)hk & m2; pr < 2u * (uint32_t)memo_cap; pr++, hi = (hi + 1u) & m2) {
          const unsigned int v = mtab[hi];
          if (v == 0u) break;
          if (mk1[v - 1] == k1 && mk2[v - 1] == k2 && mhh[v - 1] == hh) {
            sllk[rep] = mllk[v - 1];
            hit = true;
            break;
          }
        }
      }
      if (!hit && probe) {
        uint64_t tag = (skey[rep] << 1) | 1ull | P.cache_epoch;  // (the call's epoch in the upper bits of a packed genotype's tag: tag_of)
        uint64_t key = tag >> 1;
        GWords<KT> pwv = g;
        if constexpr (WIDE) {
          // the proposal's words; its tag is a hash (tag_of), a tag match is verified against the words kept beside the entry
          const int hh_ = (int)shh[rep];
          set_word<KT>(pwv, hh_ >> 4, key);
          if ((hh_ & 15) != 15) set_word<KT>(pwv, hh_ & 15, skey2[rep]);
          tag = tag_of<KT>(pwv, key_bits);
          key = tag >> 1;
        }
        uint32_t hsh = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
        hsh ^= hsh >> 16;
        hsh *= 0x7FEB352Du;
        hsh ^= hsh >> 15;
        hsh *= 0x846CA68Bu;
        hsh ^= hsh >> 16;
        const size_t set_i = (size_t)((hsh >> 12) & cache_mask);
        const ulonglong2 *set = cache + 8 * set_i;
#pragma unroll
        for (int w = 0; w < 8; w++) {
          const ulonglong2 en = set[w];
          if (en.x == tag) {
            bool same = true;
            if constexpr (WIDE) {
              const uint64_t *kw = ckeys + (8 * set_i + w) * (size_t)D.cache_key_words;
#pragma unroll
              for (int h = 0; h < KT; h++) same = same && kw[h] == pwv.w[h];
            }
            if (same) {
              sllk[rep] = __longlong_as_double((long long)en.y);
              hit = true;
            }
          }
        }
      }
      if (!hit) umiss[atomicAdd(&scal[11], 1)] = (uint16_t)rep;
    }
    __syncthreads();
    const int n_miss = scal[11];

    // ---- (3) the remaining requests, one wavefront each ----
    for (int m = wv; m < n_miss; m += FILLW_NW) {
      const int rep = (int)umiss[m];
      const uint64_t key = skey[rep];
      // which haplotypes differ from the current genotype (at most two: a dosage option replaces one word, a
      // recombination swaps segments of two), and their words
      int h1 = 0, h2 = -1, ndiff = 0;
      uint64_t w1 = g.w[0], w2 = 0;
      if constexpr (WIDE) {
        const int hh_ = (int)shh[rep];
        h1 = hh_ >> 4;
        w1 = key;
        if ((hh_ & 15) != 15) {
          h2 = hh_ & 15;
          w2 = skey2[rep];
        }
      } else {
#pragma unroll
        for (int h = 0; h < KT; h++) {
          const uint64_t wh = (key >> (key_bits * (KT - 1 - h))) & key_mask;
          if (wh != g.w[h]) {
            if (ndiff == 0) {
              h1 = h;
              w1 = wh;
            } else {
              h2 = h;
              w2 = wh;
            }
            ndiff++;
          }
        }
      }
      h1 = __builtin_amdgcn_readfirstlane(h1);
      h2 = __builtin_amdgcn_readfirstlane(h2);
      double s = 0.0;
      s += fillw_blocks<KT, true, DEEP>(FU, g, h1, w1, h2, w2, 0, nb0, lane);
      for (int cb = 4; cb < FU.nch; cb += 4) s += fillw_blocks<KT, false, DEEP>(FU, g, h1, w1, h2, w2, cb, FU.nch - cb, lane);
      const double val = wave_sum(s);
      if (lane == 0) sllk[rep] = val;
    }
    __syncthreads();
    if (memo_cap > 0) {  // what this chunk evaluated, for the chain's later chunks (while the memo has room)
      for (int m = tid; m < n_miss; m += FILLW_NT) {
        const int rep = (int)umiss[m];
        const int idx = atomicAdd(&scal[12], 1);
        if (idx < memo_cap) {
          const uint64_t k1 = skey[rep], k2 = WIDE ? skey2[rep] : 0ull;
          const uint8_t hh = WIDE ? shh[rep] : (uint8_t)0;
          mk1[idx] = k1;
          mk2[idx] = k2;
          mhh[idx] = hh;
          mllk[idx] = sllk[rep];
          const uint64_t hk = mix64(k1 ^ mix64(k2 + hh));
          const uint32_t m2 = (uint32_t)(2 * memo_cap - 1);
          for (uint32_t pr = 0, hi = (uint32_t)hk & m2; pr < 2u * (uint32_t)memo_cap; pr++, hi = (hi + 1u) & m2)
            if (mtab[hi] == 0u && atomicCAS(&mtab[hi], 0u, (unsigned int)(idx + 1)) == 0u) break;
        }
      }
      __syncthreads();
    }

    // ---- (4) option probabilities, then the totals a visit without a move would have formed ----
    // (ptab shares the keys' array: every request has been evaluated, nothing reads a key any more)
    for (int s = tid; s < n_slots; s += FILLW_NT) {
      uint32_t oin, lo;
      int my_no, ty_s;
      uint64_t min_;
      slot_option(s, oin, lo, my_no, ty_s, min_);
      const double llk_i = sllk[slrep[s]];
      double lprior_ratio = 0.0;
      if (!isnan(inbreeding)) lprior_ratio = prior_of<KT>(pt, inbreeding, dosage_of_labels(oin, lo, KT, true)) - lprior_cur;
      const int n_return = ty_s == 0 ? recombination_n_options(oin, lo, KT) : dosage_n_options(oin, lo, KT);
      const double lproposal_ratio = lninv[n_return] - lninv[my_no];
      const double mh = ((llk_i - cur_llk) + lprior_ratio) * temp + lproposal_ratio;
      ptab[s] = exp(fmin(0.0, mh) - ln[my_no]);
    }
    __syncthreads();
    if (listed) {
      double cacc = 0.0;
      for (int o = 0; o < no; o++) cacc += ptab[excl + o];
      mt[e] = cacc;
    }
    __syncthreads();
  }
}

}  // namespace mchap
