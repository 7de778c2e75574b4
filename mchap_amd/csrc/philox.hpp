// Philox4x32-10 counter-based streams for the sampler kernels (gfx950).
//
// Stream contract (DESIGN.md "Random numbers"; the CPU checker under oracle/ restates it):
//   key     = (seed_lo, seed_hi ^ stream_id_hi)
//   counter = (block_lo, block_hi, substream, stream_id_lo),  substream = chain << 16 | slot,
//             slot = temperature index, or 0xFFFF for the chain's initial-genotype stream
//   draw n of a stream = words (0,1) of block n>>1 when n is even, words (2,3) when n is odd
//   uniform double     = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53
//   integer in [0,max] = (w0 * (max + 1)) >> 32; max == 0 consumes nothing
// The reference draws from numba's private MT19937 (mchap/jitutils.py:181-183), which no
// other program can reproduce; the order in which draws are consumed is the reference's
// (SURVEY.md Appendix A.8).
#pragma once
#include "read_log.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mchap {

// Draw numbering: the draws of MCMC step i of a stream are i * STEP_DRAWS, i * STEP_DRAWS + 1, ... in the order the
// reference consumes them (a step uses fewer than 2000), so a step's draws do not depend on what earlier steps
// consumed.  The oracle's Philox mode numbers them the same way (ORC_STEP_DRAWS).
constexpr uint64_t STEP_DRAWS = 65536;

struct Rng {
  uint32_t k0, k1, c2, c3;
  uint64_t n;       // next draw index of the current stream
  uint32_t s0, s1;  // second half of the last block
  uint32_t have;    // s0/s1 hold draw n (n odd)
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;  // one v_mad_u64_u32 each
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void rng_open(Rng &g, uint64_t seed, uint64_t stream_id, uint32_t chain, uint32_t slot,
                                         uint64_t n) {
  g.k0 = (uint32_t)seed;
  g.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(stream_id >> 32);
  g.c2 = (chain << 16) | slot;
  g.c3 = (uint32_t)stream_id;
  g.n = n;
  g.have = 0;
}

__device__ __forceinline__ void rng_words(Rng &g, uint32_t &a, uint32_t &b) {
  if ((g.n & 1) && g.have) {
    a = g.s0;
    b = g.s1;
    g.have = 0;
  } else {
    uint32_t o[4];
    const uint64_t blk = g.n >> 1;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), g.c2, g.c3, g.k0, g.k1, o);
    if (g.n & 1) {
      a = o[2];
      b = o[3];
      g.have = 0;
    } else {
      a = o[0];
      b = o[1];
      g.s0 = o[2];
      g.s1 = o[3];
      g.have = 1;
    }
  }
  g.n += 1;
}

__device__ __forceinline__ double rng_double(Rng &g) {
  uint32_t a, b;
  rng_words(g, a, b);
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ uint32_t rng_interval(Rng &g, uint32_t max) {
  if (max == 0) return 0;
  uint32_t a, b;
  rng_words(g, a, b);
  return __umulhi(a, max + 1u);
}

}  // namespace mchap
