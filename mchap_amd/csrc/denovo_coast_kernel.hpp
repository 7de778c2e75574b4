// Coasting kernel of the phased de-novo sampler (kernel 5) for MI355X (gfx950).
//
// At the reference's workloads a chain reaches its mode within a few MCMC steps and then proposes for hundreds of
// steps without ever moving (assemble/mcmc.py:286-338 keeps stepping; every step still consumes its random draws).
// Whether a compound step of an UNCHANGED genotype moves is a property of the genotype and the step's uniforms alone:
//   * mutation step (mutation.py:164-246): nothing moves iff every uniform u of its K*M sub-steps has
//     mlo <= u < mhi (the bounds denovo_spec_kernel remembers from its last full evaluation);
//   * interval step (structural.py:590-673): an interval (type, start, stop) stays iff its uniform is >= the total
//     move probability of its options, which denovo_spec_kernel<.., true> leaves in the chain's table for EVERY
//     interval (PIPE_EXPORT), no-option intervals consume no uniform.
// This kernel runs exactly those tests -- the same comparisons on the same Philox draws as the speculative sampler's
// fast paths, whose consumption of the stream it reproduces draw for draw -- and no likelihood code at all.  The
// draws of MCMC step i are numbered from i * STEP_DRAWS (philox.hpp), so the steps of a settled chain are independent
// of each other: one wavefront per chain, ONE LANE PER STEP, 64 steps at a time; the first step that cannot be
// decided ends the sweep.  A chain whose step cannot be decided that way (a move, an unknown table entry, the reference's
// "breaks" error) is handed back at the START of that compound step: its record gets the step and the draw counter,
// its index is appended to P.pipe_out, and denovo_spec_kernel resumes it.  The genotype cannot change here, so the
// chain's trace rows are written in one coalesced sweep at the end.
//
// (Round 4, measured and not kept: parking the first 12 Philox blocks behind the mutation step's uniforms in LDS, computed by all
// lanes together, instead of computing the structural part's blocks where a lane needs them -- the launch over every chain went
// from 1.97 to 2.62 ms at configs[1]: the divergent blocks cost about ten block-times per sweep, fewer than the parked ones.
// What is kept of it: the block that holds the last uniform also holds the structural part's first draw and stays in registers.)
//
// Launch: workgroups of NW wavefronts -- one for the launch over every chain (grid = chains); COAST_NW_LIST for a list of
// handed-back chains, which are few (23 of 20 000 at configs[1]): NW x 64 steps per sweep, 0.25 -> 0.08 ms per launch --;
// dynamic LDS coast_lds_bytes(Mmax).
#pragma once
#include "denovo_spec_kernel.hpp"

namespace mchap {

// one chain of the launch's list (entry `slot`), by the whole workgroup
template <int NW>
__device__ __forceinline__ void coast_chain(const SimtParams &P, unsigned char *smem, int slot) {
  const DenovoParams &D = P.d;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wv = tid / WAVE;
  constexpr int NT = NW * WAVE;
  auto block_sync = [] {
    if constexpr (NW == 1) lds_sync();
    else __syncthreads();
  };
  const int Cn = D.chains, Sn = D.steps, mmax = P.max_pos;
  const int E = spec_memo_entries(mmax);
  const long long q = P.pipe_list ? (long long)P.pipe_list[slot] : (long long)slot;
  PipeState *st = reinterpret_cast<PipeState *>(P.pipe_state) + q;
  const int base = st->step;
  if (base >= Sn) return;  // finished, or stopped by an error status
  const int u = (int)(q / Cn), chain = (int)(q % Cn);
  const mchap_unit U = D.units[u];
  const int32_t *mi = P.meta_i + (size_t)u * meta_i_stride(P.max_pos);
  const int K = U.ploidy;
  const int Mh = mi[META_I_MH];
  const int n = K * Mh;
  if (st->mvalid == 0 || n > spec_draws(K, mmax)) {
    // not settled (the record's mutation bounds are not valid): straight back to the speculative sampler
    if (tid == 0) P.pipe_out[atomicAdd(P.pipe_out_count, 1)] = (int32_t)q;
    return;
  }
  LDSP(double) mt = lds_cast<double>(smem);  // [2][E] totals of the chain's interval steps
  LDSP(double) bc = mt + 2 * E;              // [Mmax] cumulative break-count distribution
  LDSP(uint64_t) tw = lds_cast<uint64_t>(smem + (size_t)8 * (2 * E + mmax));  // [K] sorted words
  LDSP(int) wfail = lds_cast<int>(smem + (size_t)8 * (2 * E + mmax + 8));      // [2][NW] first undecided step of a wavefront's sweep
  const double mlo = st->mlo, mhi = st->mhi;
  {
    const double *pm = P.pipe_memo + (size_t)q * 2 * E;
    for (int i = tid; i < 2 * E; i += NT) mt[i] = pm[i];
  }
  if (tid == 0) {
    if (D.n_intervals == 0) {
      // cumulative break-count distribution, summed in the reference's order (structural.py:44-49)
      double cacc = 0.0;
      for (int j = 0; j < Mh; j++) {
        cacc += D.break_table[(size_t)Mh * D.max_pos + j];
        bc[j] = cacc;
      }
    }
    // the trace rows of this launch: the genotype's words in ascending order
    uint64_t w[8];
#pragma unroll
    for (int h = 0; h < 8; h++) w[h] = h < K ? st->g[h] : ~0ull;
#pragma unroll
    for (int pass = 0; pass < 8; pass++) {
#pragma unroll
      for (int i = pass & 1; i + 1 < 8; i += 2) {
        const uint64_t a = w[i], b = w[i + 1];
        w[i] = a < b ? a : b;
        w[i + 1] = a < b ? b : a;
      }
    }
#pragma unroll
    for (int h = 0; h < 8; h++) tw[h] = w[h];
  }
  block_sync();
  Stream s;
  s.k0 = (uint32_t)D.seed;
  s.k1 = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
  s.c2 = ((uint32_t)chain << 16) | 0u;
  s.c3 = (uint32_t)U.stream_id;
  int stop = Sn;  // first step that cannot be decided here
  int sweep = 0;
  for (int s0 = base; s0 < Sn; s0 += NT, sweep++) {
    const int step = s0 + tid;
    bool fail = false;
    if (step < Sn) {
      const uint64_t ctr0 = (uint64_t)step * STEP_DRAWS;  // the step's draws (philox.hpp)
      // ---- mutation step: its K*M uniforms are draws ctr0 + (n-1) .. ctr0 + 2n - 2
      const uint64_t ub = ctr0 + (uint64_t)(n - 1), ue = ub + (uint64_t)n;
      // the lane's last Philox block (two draws)
      uint64_t dblk = ~0ull, dw0 = 0, dw1 = 0;
      auto words_at = [&](uint64_t d) -> uint64_t {
        const uint64_t b = d >> 1;
        if (b != dblk) {
          uint32_t o[4];
          philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), s.c2, s.c3, s.k0, s.k1, o);
          dw0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
          dw1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
          dblk = b;
        }
        return (d & 1) ? dw1 : dw0;
      };
      bool ok = true;
      for (uint64_t b = ub >> 1; 2 * b < ue; b++) {
        uint32_t o[4];
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), s.c2, s.c3, s.k0, s.k1, o);
        const uint64_t d0 = 2 * b;
        const uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32), w1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
        if (d0 >= ub && d0 < ue) {
          const double x = draw_double(w0);
          ok = ok && (mlo <= x) && (x < mhi);
        }
        if (d0 + 1 >= ub && d0 + 1 < ue) {
          const double x = draw_double(w1);
          ok = ok && (mlo <= x) && (x < mhi);
        }
        dblk = b;  // (the last block also holds draw ue, the structural part's first, when ue is odd)
        dw0 = w0;
        dw1 = w1;
      }
      fail = !ok;
      uint64_t ctr = ue;
      // ---- the three structural steps
      for (int kind = 0; kind < 3 && !fail; kind++) {
        const double pstep = kind == 0 ? D.p_recomb : (kind == 1 ? D.p_partial : D.p_dosage);
        const bool doit = draw_double(words_at(ctr)) <= pstep;
        ctr++;
        if (!doit) continue;
        uint64_t zeros;
        int n_int;
        if (kind < 2) {
          int nb;
          if (D.n_intervals > 0) {
            ctr++;  // break_dist = [0,...,0,1]: the draw is consumed (assemble/mcmc.py:214-217)
            nb = D.n_intervals - 1;
          } else {
            const double x = draw_double(words_at(ctr));
            ctr++;
            nb = Mh;
            for (int j = Mh - 1; j >= 0; j--)
              if (bc[j] > x) nb = j;  // first j with cumsum[j] > x
          }
          if (nb >= Mh) {  // the reference's "breaks" error: the speculative sampler reports it
            fail = true;
            break;
          }
          uint64_t ind = ((1ull << Mh) - 1ull) & ~1ull;
          for (int b = 0; b < nb; b++) {
            const int no = __popcll(ind);
            if (no == 0) break;
            int k = 0;
            if (no > 1) {
              k = (int)draw_interval(words_at(ctr), (uint32_t)(no - 1));
              ctr++;
            }
            uint64_t t = ind;
            while (k-- > 0) t &= t - 1;
            ind &= ~(t & (~t + 1));
          }
          zeros = ~ind & ((1ull << (Mh + 1)) - 1ull);
          n_int = nb + 1;
        } else {
          zeros = 1ull | (1ull << Mh);
          n_int = 1;
        }
        LDSP(double) mtot = mt + (kind == 0 ? 0 : E);
        if (n_int == 1) {
          const double tot = mtot[spec_memo_index(0, Mh)];
          if (tot < 0.0) continue;  // no options: no draw
          if (!(draw_double(words_at(ctr)) >= tot)) {  // a move, or an entry that is not known (NaN)
            fail = true;
            break;
          }
          ctr++;
        } else {
          // all intervals known: n_int - 1 shuffle draws, then one uniform per interval with options; nothing moves
          // if all of those are >= the largest total, whichever interval each is paired with
          double mx = -1.0;
          int n_cons = 0;
          bool unknown = false;
          uint64_t z = zeros;
          int start = __ffsll((long long)z) - 1;
          z &= z - 1;
          for (int i = 0; i < n_int; i++) {
            const int stop_ = __ffsll((long long)z) - 1;
            z &= z - 1;
            const double tot = mtot[spec_memo_index(start, stop_)];
            if (isnan(tot)) unknown = true;
            else if (tot >= 0.0) {
              n_cons++;
              mx = fmax(mx, tot);
            }
            start = stop_;
          }
          bool low = unknown;
          for (int k = 0; k < n_cons && !low; k++)
            low = !(draw_double(words_at(ctr + (uint64_t)(n_int - 1 + k))) >= mx);
          if (low) {
            fail = true;
            break;
          }
          ctr += (uint64_t)(n_int - 1 + n_cons);
        }
      }
    }
    const unsigned long long fb = __ballot(fail);
    if constexpr (NW == 1) {
      if (fb) {
        stop = s0 + __ffsll((long long)fb) - 1;  // handed back at the start of this step
        break;
      }
    } else {
      // the first undecided step over the workgroup's wavefronts (two sets of slots: a wavefront may be a sweep ahead)
      LDSP(int) wf = wfail + (sweep & 1) * NW;
      if (lane == 0) wf[wv] = fb ? s0 + wv * WAVE + __ffsll((long long)fb) - 1 : Sn;
      __syncthreads();
      int first = Sn;
#pragma unroll
      for (int w = 0; w < NW; w++) first = min(first, wf[w]);
      if (first < Sn) {
        stop = first;
        break;
      }
    }
  }
  // ---- record, hand-back list, trace rows base .. stop - 1
  if (tid == 0) {
    st->step = stop;
    if (stop < Sn) P.pipe_out[atomicAdd(P.pipe_out_count, 1)] = (int32_t)q;
  }
  const int rows = stop - base;
  uint64_t *tp = D.trace + U.trace_off + (size_t)chain * D.steps * K + (size_t)base * K;
  uint64_t *lp = reinterpret_cast<uint64_t *>(D.llks + U.llk_off + (size_t)chain * D.steps) + base;
  const int nw = rows * K;
  int h = tid % K;
  const int hs = NT % K;
  for (int i = tid; i < nw; i += NT) {
    tp[i] = tw[h];
    h += hs;
    if (h >= K) h -= K;
  }
  const uint64_t lb = (uint64_t)__double_as_longlong(st->llk);
  for (int i = tid; i < rows; i += NT) lp[i] = lb;
}

// NW == 1: a workgroup (one wavefront) per chain, the grid sized for every chain.  NW > 1: the workgroups take the entries of the
// list in turn (the list's length is only known on the device, and a grid of one 256-thread workgroup with its parked draws per
// possible entry would spend its time starting workgroups that find nothing to do).
template <int NW>
__global__ __launch_bounds__(64 * NW) void denovo_coast_kernel(const SimtParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int n_list = P.pipe_count ? *P.pipe_count : (int)((long long)P.n_units * P.d.chains);
  if constexpr (NW == 1) {
    if ((int)blockIdx.x < n_list) coast_chain<1>(P, smem, (int)blockIdx.x);
  } else {
    for (int slot = (int)blockIdx.x; slot < n_list; slot += (int)gridDim.x) {
      coast_chain<NW>(P, smem, slot);
      __syncthreads();  // (the next chain's tables overwrite this one's)
    }
  }
}

}  // namespace mchap
