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
// Launch: grid = chains, workgroups of one wavefront; dynamic LDS coast_lds_bytes().
#pragma once
#include "denovo_spec_kernel.hpp"

namespace mchap {

__global__ __launch_bounds__(64) void denovo_coast_kernel(const SimtParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenovoParams &D = P.d;
  const int lane = threadIdx.x;
  const int Cn = D.chains, Sn = D.steps, mmax = P.max_pos;
  const int E = spec_memo_entries(mmax);
  const long long n_chains = (long long)P.n_units * Cn;
  const int n_list = P.pipe_count ? *P.pipe_count : (int)n_chains;
  if ((long long)blockIdx.x >= n_list) return;  // the grid is sized for every chain
  const long long q = P.pipe_list ? (long long)P.pipe_list[blockIdx.x] : (long long)blockIdx.x;
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
    if (lane == 0) P.pipe_out[atomicAdd(P.pipe_out_count, 1)] = (int32_t)q;
    return;
  }
  LDSP(double) mt = lds_cast<double>(smem);  // [2][E] totals of the chain's interval steps
  LDSP(double) bc = mt + 2 * E;              // [Mmax] cumulative break-count distribution
  LDSP(uint64_t) tw = lds_cast<uint64_t>(smem + (size_t)8 * (2 * E + mmax));  // [K] sorted words
  const double mlo = st->mlo, mhi = st->mhi;
  {
    const double *pm = P.pipe_memo + (size_t)q * 2 * E;
    for (int i = lane; i < 2 * E; i += WAVE) mt[i] = pm[i];
  }
  if (lane == 0) {
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
  lds_sync();
  Stream s;
  s.k0 = (uint32_t)D.seed;
  s.k1 = (uint32_t)(D.seed >> 32) ^ (uint32_t)(U.stream_id >> 32);
  s.c2 = ((uint32_t)chain << 16) | 0u;
  s.c3 = (uint32_t)U.stream_id;
  int stop = Sn;  // first step that cannot be decided here
  for (int s0 = base; s0 < Sn; s0 += WAVE) {
    const int step = s0 + lane;
    bool fail = false;
    if (step < Sn) {
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
      const uint64_t ctr0 = (uint64_t)step * STEP_DRAWS;  // the step's draws (philox.hpp)
      // ---- mutation step: its K*M uniforms are draws ctr0 + (n-1) .. ctr0 + 2n - 2
      const uint64_t ub = ctr0 + (uint64_t)(n - 1), ue = ub + (uint64_t)n;
      bool ok = true;
      for (uint64_t b = ub >> 1; 2 * b < ue; b++) {
        uint32_t o[4];
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), s.c2, s.c3, s.k0, s.k1, o);
        const uint64_t d0 = 2 * b;
        if (d0 >= ub) {
          const double x = draw_double((uint64_t)o[0] | ((uint64_t)o[1] << 32));
          ok = ok && (mlo <= x) && (x < mhi);
        }
        if (d0 + 1 < ue) {
          const double x = draw_double((uint64_t)o[2] | ((uint64_t)o[3] << 32));
          ok = ok && (mlo <= x) && (x < mhi);
        }
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
    if (fb) {
      stop = s0 + __ffsll((long long)fb) - 1;  // handed back at the start of this step
      break;
    }
  }
  // ---- record, hand-back list, trace rows base .. stop - 1
  if (lane == 0) {
    st->step = stop;
    if (stop < Sn) P.pipe_out[atomicAdd(P.pipe_out_count, 1)] = (int32_t)q;
  }
  const int rows = stop - base;
  uint64_t *tp = D.trace + U.trace_off + (size_t)chain * D.steps * K + (size_t)base * K;
  uint64_t *lp = reinterpret_cast<uint64_t *>(D.llks + U.llk_off + (size_t)chain * D.steps) + base;
  const int nw = rows * K;
  int h = lane % K;
  const int hs = WAVE % K;
  for (int i = lane; i < nw; i += WAVE) {
    tp[i] = tw[h];
    h += hs;
    if (h >= K) h -= K;
  }
  const uint64_t lb = (uint64_t)__double_as_longlong(st->llk);
  for (int i = lane; i < rows; i += WAVE) lp[i] = lb;
}

}  // namespace mchap
