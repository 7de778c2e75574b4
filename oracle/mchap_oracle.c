/*
 * mchap_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's per-locus MCMC haplotype assembler and exact
 * genotype caller (PlantandFoodResearch/MCHap v0.11.1).  See mchap_oracle.h for the
 * rules on who may load this library.  Arithmetic is IEEE fp64 in the reference's own
 * sequential order (build with -ffp-contract=off, no fast-math).  All reference
 * citations are relative to /root/reference/mchap/.
 *
 * Random numbers.  The reference draws from numba's private MT19937 stream, which
 * cannot be reproduced outside numba.  Two generators are provided:
 *   ORC_RNG_NUMPY_MT19937  numpy's legacy RandomState algorithms (random_sample,
 *        masked-rejection bounded ints, Fisher-Yates shuffle/permutation, choice), i.e.
 *        exactly what the reference consumes when imported with an identity-njit shim.
 *        Whole traces of this oracle are compared step-for-step with traces captured
 *        from the reference in that mode (tests/golden).
 *   ORC_RNG_PHILOX  the counter-based Philox4x32-10 streams the HIP kernels use
 *        (same order of consumption, one stream per (unit, chain, temperature); the draws
 *        of MCMC step i are numbered from i * ORC_STEP_DRAWS, so a step's draws do not depend on
 *        what earlier steps consumed), so that kernel traces can be compared step-for-step
 *        with this oracle.
 */
#define _DEFAULT_SOURCE /* lgamma_r */
#include "mchap_oracle.h"

#include <math.h>
#include <stdlib.h>
#ifdef __GLIBC__
#include <malloc.h>
#endif
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* glibc's lgamma() stores the sign in the global `signgam`: sixteen threads of the CPU baseline writing that one cache line made
   orc_call_mcmc scale 1.5-fold on 16 threads.  lgamma_r: the same values, the sign in a local. */
static double orc_lgamma(double x) {
  int sign;
  return lgamma_r(x, &sign);
}
#define lgamma(x) orc_lgamma(x)

const char *orc_version(void) { return "mchap-oracle 0.1 (restates MCHap v0.11.1)"; }

/* ------------------------------------------------------------------------- */
/* RNG                                                                        */
/* ------------------------------------------------------------------------- */

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* draw n of stream (seed, stream_id, substream): half (n&1) of philox block n>>1 */
static void philox_words(uint64_t seed, uint64_t stream_id, uint32_t substream, uint64_t n, uint32_t *w0, uint32_t *w1) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(stream_id >> 32)};
  uint64_t blk = n >> 1;
  uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), substream, (uint32_t)stream_id};
  uint32_t out[4];
  orc_philox4x32_10(ctr, key, out);
  if (n & 1) { *w0 = out[2]; *w1 = out[3]; } else { *w0 = out[0]; *w1 = out[1]; }
}

static double words_to_double(uint32_t a, uint32_t b) {
  /* 53-bit construction shared with numpy's legacy random_sample */
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

double orc_philox_double(uint64_t seed, uint64_t stream_id, uint32_t substream, uint64_t n) {
  uint32_t a, b;
  philox_words(seed, stream_id, substream, n, &a, &b);
  return words_to_double(a, b);
}

uint32_t orc_philox_interval(uint64_t seed, uint64_t stream_id, uint32_t substream, uint64_t n, uint32_t max) {
  uint32_t a, b;
  philox_words(seed, stream_id, substream, n, &a, &b);
  return (uint32_t)(((uint64_t)a * ((uint64_t)max + 1)) >> 32);
}

typedef struct {
  uint32_t mt[624];
  int pos;
} mt_state;

static void mt_seed(mt_state *s, uint32_t seed) {
  /* numpy legacy seeding for an integer seed (init_genrand) */
  for (int i = 0; i < 624; i++) {
    s->mt[i] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
  }
  s->pos = 624;
}

static uint32_t mt_next32(mt_state *s) {
  if (s->pos >= 624) {
    uint32_t *mt = s->mt;
    int i;
    for (i = 0; i < 624 - 397; i++) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; i++) {
      uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
      mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    s->pos = 0;
  }
  uint32_t y = s->mt[s->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

#define ORC_SUB_INIT 0xFFFFu
#define ORC_STEP_DRAWS 65536u /* draw numbers reserved per MCMC step and stream (Philox mode); a step uses < 2000 */

typedef struct {
  int kind;
  /* philox */
  uint64_t seed, stream_id;
  uint32_t chain;
  uint64_t n[ORC_MAX_TEMPS + 1]; /* draw counters; slot ORC_MAX_TEMPS == init stream */
  /* numpy mt */
  mt_state *mt; /* shared across chains (the reference seeds once per fit) */
} orc_rng;

static inline uint32_t rng_substream(const orc_rng *g, int slot) {
  uint32_t t = (slot == ORC_MAX_TEMPS) ? ORC_SUB_INIT : (uint32_t)slot;
  return (g->chain << 16) | t;
}

/* uniform double in [0,1): np.random.rand()/random() */
static double rng_double(orc_rng *g, int slot) {
  if (g->kind == ORC_RNG_NUMPY_MT19937) {
    uint32_t a = mt_next32(g->mt), b = mt_next32(g->mt);
    return words_to_double(a, b);
  }
  return orc_philox_double(g->seed, g->stream_id, rng_substream(g, slot), g->n[slot]++);
}

/* uniform integer in [0, max]: numpy random_interval / bounded masked uint32 */
static uint32_t rng_interval(orc_rng *g, int slot, uint32_t max) {
  if (max == 0) return 0; /* numpy consumes nothing; philox mode follows suit */
  if (g->kind == ORC_RNG_NUMPY_MT19937) {
    uint32_t mask = max, v;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((v = (mt_next32(g->mt) & mask)) > max) {}
    return v;
  }
  return orc_philox_interval(g->seed, g->stream_id, rng_substream(g, slot), g->n[slot]++, max);
}

void orc_mt_doubles(uint32_t seed, int n, double *out) {
  mt_state s;
  mt_seed(&s, seed);
  for (int i = 0; i < n; i++) {
    uint32_t a = mt_next32(&s), b = mt_next32(&s);
    out[i] = words_to_double(a, b);
  }
}

/* jitutils.py:77-92 random_choice: searchsorted(cumsum(p), u, side="right").
 * A result of n (u beyond the last cumulative value, probability ~1e-16) is returned as is. */
static int choose_from(const double *p, int n, double u) {
  double c = 0.0;
  for (int i = 0; i < n; i++) {
    c += p[i];
    if (c > u) return i;
  }
  return n;
}

/* ------------------------------------------------------------------------- */
/* jitutils                                                                   */
/* ------------------------------------------------------------------------- */

/* jitutils.py:7-26 */
double orc_add_log_prob(double x, double y) {
  if (x == -INFINITY && y == -INFINITY) return -INFINITY;
  if (x > y) return x + log1p(exp(y - x));
  return y + log1p(exp(x - y));
}

/* jitutils.py:51-74 (sum_log_probs 30-47) */
static void normalise_log_probs(const double *llks, int n, double *out) {
  double acc = llks[0];
  for (int i = 1; i < n; i++) acc = orc_add_log_prob(acc, llks[i]);
  for (int i = 0; i < n; i++) out[i] = exp(llks[i] - acc);
}

/* jitutils.py:114-146 */
void orc_increment_genotype(int64_t *g, int ploidy) {
  if (ploidy == 1) { g[0] += 1; return; }
  int64_t previous = g[0];
  for (int i = 1; i < ploidy; i++) {
    int64_t allele = g[i];
    if (allele == previous) continue;
    if (allele > previous) {
      int k = i - 1;
      g[k] += 1;
      for (int z = 0; z < k; z++) g[z] = 0;
      return;
    }
    return; /* not ascending: reference raises */
  }
  g[ploidy - 1] += 1;
  for (int z = 0; z < ploidy - 1; z++) g[z] = 0;
}

/* jitutils.py:195-210 _comb */
static int64_t gcd64(int64_t x, int64_t y) {
  while (y != 0) { int64_t t = x % y; x = y; y = t; }
  return x;
}
static int64_t comb64(int64_t n, int64_t k) {
  if (n < 0 || k < 0) return -1;
  if (k > n) return 0;
  int64_t r = 1;
  for (int64_t d = 1; d <= k; d++) {
    int64_t g = gcd64(r, d);
    r /= g;
    r *= n;
    r /= d / g;
    n -= 1;
  }
  return r;
}

/* jitutils.py:228-250 */
int64_t orc_comb_with_replacement(int64_t n, int64_t k) {
  if (n == 0 && k == 0) return 0;
  return comb64(n + k - 1, k);
}

/* jitutils.py:253-276 */
int64_t orc_genotype_alleles_as_index(const int64_t *alleles, int ploidy) {
  int64_t index = 0;
  for (int i = 0; i < ploidy; i++) {
    int64_t a = alleles[i];
    if (a >= 0) index += orc_comb_with_replacement(a, i + 1);
  }
  return index;
}

/* jitutils.py:279-318 */
void orc_index_as_genotype_alleles(int64_t index, int ploidy, int64_t *out) {
  if (index < 0) { for (int i = 0; i < ploidy; i++) out[i] = -1; return; }
  int64_t remainder = index;
  for (int idx = 0; idx < ploidy; idx++) {
    int p = ploidy - idx;
    int64_t n = -1, nw = 0, prev = 0;
    while (nw <= remainder) {
      n += 1;
      prev = nw;
      nw = orc_comb_with_replacement(n, p);
    }
    n -= 1;
    remainder -= prev;
    out[p - 1] = n;
  }
}

/* jitutils.py:149-171 */
static double ln_equivalent_permutations_i8(const int8_t *dosage, int n) {
  int ploidy = 0;
  for (int i = 0; i < n; i++) ploidy += dosage[i];
  double num = lgamma((double)ploidy + 1.0);
  double den = 0.0;
  for (int i = 0; i < n; i++) den += lgamma((double)dosage[i] + 1.0);
  return num - den;
}

/* jitutils.py:321-347 array_equal over a row of width n */
static int rows_equal(const int8_t *x, const int8_t *y, int n) {
  for (int i = 0; i < n; i++) if (x[i] != y[i]) return 0;
  return 1;
}

/* jitutils.py:350-375 */
int orc_count_haplotype_copies(const int8_t *genotype, int ploidy, int n_base, int h) {
  int count = 1;
  for (int i = 0; i < ploidy; i++) {
    if (i == h) continue;
    if (rows_equal(genotype + (size_t)i * n_base, genotype + (size_t)h * n_base, n_base)) count++;
  }
  return count;
}

/* jitutils.py:378-422; `stride` lets the caller pass a column slice (labels[:, 0:1]) */
static void haplotype_dosage_strided(int8_t *dosage, const int8_t *rows, int ploidy, int width, int stride) {
  for (int h = 0; h < ploidy; h++) dosage[h] = 1;
  for (int h = 0; h < ploidy; h++) {
    if (dosage[h] == 0) continue;
    for (int p = h + 1; p < ploidy; p++) {
      if (dosage[p] == 0) continue;
      if (rows_equal(rows + (size_t)h * stride, rows + (size_t)p * stride, width)) {
        dosage[h] += 1;
        dosage[p] = 0;
      }
    }
  }
}
void orc_get_haplotype_dosage(int8_t *dosage, const int8_t *genotype, int ploidy, int n_base) {
  haplotype_dosage_strided(dosage, genotype, ploidy, n_base, n_base);
}

/* jitutils.py:501-544 */
void orc_structural_change(int8_t *genotype, int ploidy, int n_base, const int8_t *idx, int start, int stop) {
  int8_t cache[ORC_MAX_PLOIDY];
  for (int j = start; j < stop; j++) {
    for (int h = 0; h < ploidy; h++) cache[h] = genotype[(size_t)h * n_base + j];
    for (int h = 0; h < ploidy; h++) genotype[(size_t)h * n_base + j] = cache[idx[h]];
  }
}

/* ------------------------------------------------------------------------- */
/* likelihood (assemble/likelihood.py)                                        */
/* ------------------------------------------------------------------------- */

/* likelihood.py:17-70 */
double orc_log_likelihood(const double *reads, int n_reads, int n_pos, int max_allele,
                          const int8_t *genotype, int ploidy, const int64_t *read_counts) {
  double llk = 0.0;
  for (int r = 0; r < n_reads; r++) {
    double read_prob = 0.0;
    const double *rd = reads + (size_t)r * n_pos * max_allele;
    for (int h = 0; h < ploidy; h++) {
      double prod = 1.0;
      const int8_t *hap = genotype + (size_t)h * n_pos;
      for (int j = 0; j < n_pos; j++) {
        double val = rd[(size_t)j * max_allele + hap[j]];
        if (!isnan(val)) prod *= val;
      }
      read_prob += prod / (double)ploidy;
    }
    double lrp = log(read_prob);
    if (read_counts) lrp *= (double)read_counts[r];
    llk += lrp;
  }
  return llk;
}

/* likelihood.py:73-148 */
double orc_log_likelihood_structural_change(const double *reads, int n_reads, int n_pos, int max_allele,
                                            const int8_t *genotype, int ploidy, const int8_t *hidx,
                                            int start, int stop, const int64_t *read_counts) {
  double llk = 0.0;
  for (int r = 0; r < n_reads; r++) {
    double read_prob = 0.0;
    const double *rd = reads + (size_t)r * n_pos * max_allele;
    for (int h = 0; h < ploidy; h++) {
      double prod = 1.0;
      for (int j = 0; j < n_pos; j++) {
        int h_ = (j >= start && j < stop) ? hidx[h] : h;
        double val = rd[(size_t)j * max_allele + genotype[(size_t)h_ * n_pos + j]];
        if (!isnan(val)) prod *= val;
      }
      read_prob += prod / (double)ploidy;
    }
    double lrp = log(read_prob);
    if (read_counts) lrp *= (double)read_counts[r];
    llk += lrp;
  }
  return llk;
}

/* ------------------------------------------------------------------------- */
/* llk cache: array-backed trie (assemble/arraymap.py:5-168,                  */
/* likelihood.py:151-305).  Results-neutral by construction.                  */
/* ------------------------------------------------------------------------- */

typedef struct {
  int64_t *tree;   /* [n_nodes][branches] */
  double *values;  /* [n_values] */
  int64_t n_nodes, n_values;
  int array_length, branches;
  int64_t empty_node, empty_value, max_size;
} orc_cache;

static orc_cache *cache_new(int array_length, int branches) {
  /* likelihood.py:189-191 -> arraymap.new(initial_size=64, max_size=2**16) */
  orc_cache *c = (orc_cache *)malloc(sizeof(orc_cache));
  c->array_length = array_length;
  c->branches = branches;
  c->n_nodes = 64;
  c->n_values = 64;
  c->max_size = 1 << 16;
  c->tree = (int64_t *)malloc(sizeof(int64_t) * c->n_nodes * branches);
  c->values = (double *)malloc(sizeof(double) * c->n_values);
  for (int64_t i = 0; i < c->n_nodes * branches; i++) c->tree[i] = -1;
  for (int64_t i = 0; i < c->n_values; i++) c->values[i] = NAN;
  c->empty_node = 1;
  c->empty_value = 0;
  return c;
}
static void cache_free(orc_cache *c) {
  if (!c) return;
  free(c->tree);
  free(c->values);
  free(c);
}
static void cache_clear(orc_cache *c) {
  /* arraymap.py:96-98,120-122: emptied in place, sizes kept */
  for (int64_t i = 0; i < c->n_nodes * c->branches; i++) c->tree[i] = -1;
  for (int64_t i = 0; i < c->n_values; i++) c->values[i] = NAN;
  c->empty_node = 1;
  c->empty_value = 0;
}
/* arraymap.py:136-168 */
static double cache_get(const orc_cache *c, const int8_t *array) {
  int64_t node = 0;
  for (int i = 0; i < c->array_length; i++) {
    int64_t next = c->tree[node * c->branches + array[i]];
    if (next < 0) return c->values[c->empty_value]; /* NaN slot */
    node = next;
  }
  int64_t vi = c->tree[node * c->branches + 0];
  if (vi < 0) return c->values[c->empty_value];
  return c->values[vi];
}
/* arraymap.py:49-133 with empty_if_full=True */
static void cache_set(orc_cache *c, const int8_t *array, double value) {
  int64_t node = 0;
  for (int i = 0; i < c->array_length; i++) {
    int j = array[i];
    int64_t next = c->tree[node * c->branches + j];
    if (next < 0) {
      next = c->empty_node;
      c->tree[node * c->branches + j] = next;
      c->empty_node += 1;
      if (c->empty_node + 1 >= c->n_nodes) {
        if (c->n_nodes * 2 > c->max_size) { cache_clear(c); return; }
        int64_t nn = c->n_nodes * 2;
        c->tree = (int64_t *)realloc(c->tree, sizeof(int64_t) * nn * c->branches);
        for (int64_t z = c->n_nodes * c->branches; z < nn * c->branches; z++) c->tree[z] = -1;
        c->n_nodes = nn;
      }
    }
    node = next;
  }
  int64_t vi = c->tree[node * c->branches + 0];
  if (vi < 0) {
    vi = c->empty_value;
    c->tree[node * c->branches + 0] = vi;
    c->empty_value += 1;
    if (c->empty_value + 1 >= c->n_values) {
      if (c->n_values * 2 > c->max_size) { cache_clear(c); return; }
      int64_t nv = c->n_values * 2;
      c->values = (double *)realloc(c->values, sizeof(double) * nv);
      for (int64_t z = c->n_values; z < nv; z++) c->values[z] = NAN;
      c->n_values = nv;
    }
  }
  c->values[vi] = value;
}

/* sampler context shared by the step functions */
typedef struct {
  const double *reads;
  int n_reads, n_pos, max_allele;
  const int64_t *read_counts;
  const int8_t *n_alleles;
  int ploidy;
  double inbreeding; /* NaN == None */
  double log_unique_haplotypes;
  orc_cache *cache;
  orc_rng *rng;
  orc_stats *stats;
} orc_ctx;

/* Optional event log for tests/aids/analyze_moves.py (only with -DORC_MOVE_LOG, a separate build: the shipped oracle has none of it).
   Entry = (FNV-1a hash of the ordered genotype bytes [+ step type, start, stop for kind 3]) << 2 | kind:
   0 step start, 1 accepted mutation, 2 accepted structural move, 3 interval step visited (with options). */
#ifdef ORC_MOVE_LOG
static __thread uint64_t *orc_mlog_buf = NULL;
static __thread size_t orc_mlog_cap = 0, orc_mlog_n = 0;
void orc_move_log_set(uint64_t *buf, size_t cap) { orc_mlog_buf = buf; orc_mlog_cap = cap; orc_mlog_n = 0; }
size_t orc_move_log_count(void) { return orc_mlog_n; }
static void orc_mlog(const orc_ctx *c, const int8_t *genotype, int kind, int a, int b, int t) {
  if (!orc_mlog_buf) return;
  uint64_t h = 1469598103934665603ull;
  size_t n = (size_t)c->ploidy * c->n_pos;
  for (size_t i = 0; i < n; i++) { h ^= (uint8_t)genotype[i]; h *= 1099511628211ull; }
  if (kind == 3) { h ^= (uint64_t)(a * 65536 + b * 256 + t + 1); h *= 1099511628211ull; }
  if (orc_mlog_n < orc_mlog_cap) orc_mlog_buf[orc_mlog_n] = (h << 2) | (uint64_t)kind;
  orc_mlog_n++;
}
#define ORC_MLOG(c, g, k, a, b, t) orc_mlog(c, g, k, a, b, t)
#else
#define ORC_MLOG(c, g, k, a, b, t) ((void)0)
#endif

/* likelihood.py:194-235 */
static double llk_cached(orc_ctx *c, const int8_t *genotype) {
  if (!c->cache) {
    if (c->stats) c->stats->llk_evals++;
    return orc_log_likelihood(c->reads, c->n_reads, c->n_pos, c->max_allele, genotype, c->ploidy, c->read_counts);
  }
  double v = cache_get(c->cache, genotype);
  if (isnan(v)) {
    if (c->stats) c->stats->llk_evals++;
    v = orc_log_likelihood(c->reads, c->n_reads, c->n_pos, c->max_allele, genotype, c->ploidy, c->read_counts);
    cache_set(c->cache, genotype, v);
  } else if (c->stats) c->stats->llk_cache_hits++;
  return v;
}

/* likelihood.py:238-305 */
static double llk_structural_cached(orc_ctx *c, const int8_t *genotype, const int8_t *hidx, int start, int stop) {
  if (!c->cache) {
    if (c->stats) c->stats->llk_evals++;
    return orc_log_likelihood_structural_change(c->reads, c->n_reads, c->n_pos, c->max_allele, genotype, c->ploidy,
                                                hidx, start, stop, c->read_counts);
  }
  int8_t gnew[ORC_MAX_PLOIDY * ORC_MAX_POS];
  memcpy(gnew, genotype, (size_t)c->ploidy * c->n_pos);
  orc_structural_change(gnew, c->ploidy, c->n_pos, hidx, start, stop);
  double v = cache_get(c->cache, gnew);
  if (isnan(v)) {
    if (c->stats) c->stats->llk_evals++;
    v = orc_log_likelihood_structural_change(c->reads, c->n_reads, c->n_pos, c->max_allele, genotype, c->ploidy,
                                             hidx, start, stop, c->read_counts);
    cache_set(c->cache, gnew, v);
  } else if (c->stats) c->stats->llk_cache_hits++;
  return v;
}

/* ------------------------------------------------------------------------- */
/* priors                                                                     */
/* ------------------------------------------------------------------------- */

/* assemble/prior.py:15-112 */
double orc_assemble_log_genotype_prior(const int8_t *dosage, int ploidy_len, double log_unique_haplotypes, double inbreeding) {
  int ploidy = 0;
  for (int i = 0; i < ploidy_len; i++) ploidy += dosage[i];
  if (inbreeding == 0.0) {
    /* prior.py:15-36 null prior */
    double ln_perms = ln_equivalent_permutations_i8(dosage, ploidy_len);
    double ln_total = (double)ploidy * log_unique_haplotypes;
    return ln_perms - ln_total;
  }
  /* prior.py:107-112 */
  double log_dispersion = log((1.0 - inbreeding) / inbreeding) - log_unique_haplotypes;
  /* prior.py:39-78 */
  double dispersion = exp(log_dispersion);
  double sum_dispersion = exp(log_dispersion + log_unique_haplotypes);
  double num = lgamma((double)ploidy + 1.0) + lgamma(sum_dispersion);
  double den = lgamma((double)ploidy + sum_dispersion);
  double left = num - den;
  double prod = 0.0;
  for (int i = 0; i < ploidy_len; i++) {
    int dose = dosage[i];
    if (dose > 0) {
      double n2 = lgamma((double)dose + dispersion);
      double d2 = lgamma((double)dose + 1.0) + lgamma(dispersion);
      prod += n2 - d2;
    }
  }
  return left + prod;
}

/* calling/utils.py:7-35 */
static void allelic_dosage(const int64_t *g, int ploidy, int64_t *dosage) {
  for (int i = 0; i < ploidy; i++) dosage[i] = 0;
  for (int i = 0; i < ploidy; i++) {
    int j = 0;
    while (g[i] != g[j]) j++;
    dosage[j] += 1;
  }
}

/* calling/prior.py:116-179 (calculate_alphas 10-27) */
double orc_calling_log_genotype_prior(const int64_t *genotype, int ploidy, int64_t unique_haplotypes,
                                      double inbreeding, const double *frequencies) {
  int64_t dosage[ORC_MAX_PLOIDY];
  allelic_dosage(genotype, ploidy, dosage);
  if (inbreeding == 0.0) {
    /* ln_equivalent_permutations over int dosage */
    int tot = 0;
    for (int i = 0; i < ploidy; i++) tot += (int)dosage[i];
    double num = lgamma((double)tot + 1.0), den = 0.0;
    for (int i = 0; i < ploidy; i++) den += lgamma((double)dosage[i] + 1.0);
    double ln_perms = num - den;
    if (!frequencies) return ln_perms - (double)ploidy * log((double)unique_haplotypes);
    double prod = 1.0;
    for (int i = 0; i < ploidy; i++) prod *= frequencies[genotype[i]];
    return ln_perms + log(prod);
  }
  double alpha_const = 0.0, sum_alphas;
  double scale = (1.0 - inbreeding) / inbreeding;
  if (!frequencies) {
    alpha_const = (1.0 / (double)unique_haplotypes) * scale;
    sum_alphas = alpha_const * (double)unique_haplotypes;
  } else {
    /* alphas.sum(): sequential (numba) */
    sum_alphas = 0.0;
    for (int64_t i = 0; i < unique_haplotypes; i++) sum_alphas += frequencies[i] * scale;
  }
  double num = lgamma((double)ploidy + 1.0) + lgamma(sum_alphas);
  double den = lgamma((double)ploidy + sum_alphas);
  double left = num - den;
  double prod = 0.0;
  for (int i = 0; i < ploidy; i++) {
    int64_t dose = dosage[i];
    if (dose > 0) {
      double alpha_i = frequencies ? frequencies[genotype[i]] * scale : alpha_const;
      double n2 = lgamma((double)dose + alpha_i);
      double d2 = lgamma((double)dose + 1.0) + lgamma(alpha_i);
      prod += n2 - d2;
    }
  }
  return left + prod;
}

static double ctx_dosage_prior(const orc_ctx *c, const int8_t *rows, int width) {
  /* mutation.py:89-99 / structural.py:509-521: flat prior when inbreeding is None */
  if (isnan(c->inbreeding)) return 0.0;
  int8_t dosage[ORC_MAX_PLOIDY];
  haplotype_dosage_strided(dosage, rows, c->ploidy, width, width);
  return orc_assemble_log_genotype_prior(dosage, c->ploidy, c->log_unique_haplotypes, c->inbreeding);
}

/* ------------------------------------------------------------------------- */
/* mutation step (assemble/mutation.py)                                       */
/* ------------------------------------------------------------------------- */

/* mutation.py:14-161 up to (not including) random_choice.  genotype is restored. */
static void base_step_probs(orc_ctx *c, int8_t *genotype, double llk, int h, int j, double temp,
                            double *probs, double *llks) {
  int n_base = c->n_pos;
  int n_alleles = c->n_alleles[j];
  double log_accept[ORC_MAX_ALLELE];
  double lhapcount = log((double)orc_count_haplotype_copies(genotype, c->ploidy, n_base, h));
  double lprior = ctx_dosage_prior(c, genotype, n_base);
  int8_t current = genotype[(size_t)h * n_base + j];
  int n_options = 0;
  for (int i = 0; i < n_alleles; i++) {
    if (i == current) {
      llks[i] = llk;
      log_accept[i] = -INFINITY;
    } else {
      n_options += 1;
      genotype[(size_t)h * n_base + j] = (int8_t)i;
      if (c->stats) c->stats->mutation_evals++;
      double llk_i = llk_cached(c, genotype);
      llks[i] = llk_i;
      double llk_ratio = llk_i - llk;
      double lprior_ratio = 0.0;
      if (!isnan(c->inbreeding)) lprior_ratio = ctx_dosage_prior(c, genotype, n_base) - lprior;
      double lhapcount_i = log((double)orc_count_haplotype_copies(genotype, c->ploidy, n_base, h));
      double lproposal_ratio = lhapcount_i - lhapcount;
      double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
      log_accept[i] = fmin(0.0, mh);
    }
  }
  genotype[(size_t)h * n_base + j] = current;
  double ln_opt = log((double)n_options);
  double sum = 0.0;
  for (int i = 0; i < n_alleles; i++) {
    probs[i] = exp(log_accept[i] - ln_opt);
    sum += probs[i];
  }
  probs[current] = 1.0 - sum;
}

int orc_base_step_probabilities(const double *reads, int n_reads, int n_pos, int max_allele,
                                const int8_t *genotype, int ploidy, double llk, int h, int j, int n_alleles_j,
                                double log_unique_haplotypes, double inbreeding, double temp,
                                const int64_t *read_counts, double *probs, double *llks) {
  int8_t na[ORC_MAX_POS];
  int8_t g[ORC_MAX_PLOIDY * ORC_MAX_POS];
  if (n_pos > ORC_MAX_POS || ploidy > ORC_MAX_PLOIDY || n_alleles_j > ORC_MAX_ALLELE) return ORC_ERR_LIMIT;
  for (int i = 0; i < n_pos; i++) na[i] = (int8_t)n_alleles_j;
  memcpy(g, genotype, (size_t)ploidy * n_pos);
  orc_ctx c = {reads, n_reads, n_pos, max_allele, read_counts, na, ploidy, inbreeding, log_unique_haplotypes, NULL, NULL, NULL};
  base_step_probs(&c, g, llk, h, j, temp, probs, llks);
  return ORC_OK;
}

static double base_step(orc_ctx *c, int8_t *genotype, double llk, int h, int j, double temp, int slot) {
  double probs[ORC_MAX_ALLELE], llks[ORC_MAX_ALLELE];
  int n_alleles = c->n_alleles[j];
  base_step_probs(c, genotype, llk, h, j, temp, probs, llks);
  int choice = choose_from(probs, n_alleles, rng_double(c->rng, slot));
  if (choice >= n_alleles) choice = n_alleles - 1;
  int8_t before = genotype[(size_t)h * c->n_pos + j];
  genotype[(size_t)h * c->n_pos + j] = (int8_t)choice;
  if (before != (int8_t)choice) ORC_MLOG(c, genotype, 1, 0, 0, 0);
  (void)before;
  return llks[choice];
}

/* mutation.py:164-246 */
static double mutation_compound_step(orc_ctx *c, int8_t *genotype, double llk, double temp, int slot) {
  int n = c->ploidy * c->n_pos;
  int8_t sub[ORC_MAX_PLOIDY * ORC_MAX_POS][2];
  for (int h = 0; h < c->ploidy; h++)
    for (int j = 0; j < c->n_pos; j++) {
      sub[h * c->n_pos + j][0] = (int8_t)h;
      sub[h * c->n_pos + j][1] = (int8_t)j;
    }
  /* np.random.shuffle(substeps): Fisher-Yates from the top */
  for (int i = n - 1; i >= 1; i--) {
    int k = (int)rng_interval(c->rng, slot, (uint32_t)i);
    if (k == i) continue;
    int8_t t0 = sub[k][0], t1 = sub[k][1];
    sub[k][0] = sub[i][0]; sub[k][1] = sub[i][1];
    sub[i][0] = t0; sub[i][1] = t1;
  }
  for (int i = 0; i < n; i++) llk = base_step(c, genotype, llk, sub[i][0], sub[i][1], temp, slot);
  return llk;
}

/* ------------------------------------------------------------------------- */
/* structural steps (assemble/structural.py)                                  */
/* ------------------------------------------------------------------------- */

/* structural.py:310-360 over columns selected by `cols` */
static void label_haplotypes(int8_t *labels, int lstride, const int8_t *genotype, int ploidy, int n_base,
                             const int *cols, int n_cols) {
  for (int h = 0; h < ploidy; h++) labels[h * lstride] = 0;
  for (int ci = 0; ci < n_cols; ci++) {
    int i = cols[ci];
    for (int j = 1; j < ploidy; j++) {
      if (genotype[(size_t)j * n_base + i] == genotype[(size_t)labels[j * lstride] * n_base + i]) continue;
      int8_t prev = labels[j * lstride];
      labels[j * lstride] = (int8_t)j;
      for (int k = j + 1; k < ploidy; k++) {
        if (labels[k * lstride] == prev && genotype[(size_t)j * n_base + i] == genotype[(size_t)k * n_base + i])
          labels[k * lstride] = (int8_t)j;
      }
    }
  }
}

/* structural.py:393-430 */
void orc_haplotype_segment_labels(const int8_t *genotype, int ploidy, int n_base, int start, int stop, int8_t *labels) {
  int cols[ORC_MAX_POS], n = 0;
  for (int j = start; j < stop; j++) cols[n++] = j;
  label_haplotypes(labels + 0, 2, genotype, ploidy, n_base, cols, n);
  n = 0;
  for (int j = 0; j < n_base; j++) if (j < start || j >= stop) cols[n++] = j;
  label_haplotypes(labels + 1, 2, genotype, ploidy, n_base, cols, n);
}

/* structural.py:74-118 */
int orc_recombination_step_n_options(const int8_t *labels, int ploidy) {
  int8_t dosage[ORC_MAX_PLOIDY];
  haplotype_dosage_strided(dosage, labels, ploidy, 2, 2);
  int n = 0;
  for (int h0 = 0; h0 < ploidy; h0++) {
    if (dosage[h0] == 0) continue;
    for (int h1 = h0 + 1; h1 < ploidy; h1++) {
      if (dosage[h1] == 0) continue;
      if (labels[h0 * 2] == labels[h1 * 2] || labels[h0 * 2 + 1] == labels[h1 * 2 + 1]) continue;
      n++;
    }
  }
  return n;
}

/* structural.py:121-178 */
int orc_recombination_step_options(const int8_t *labels, int ploidy, int8_t *options) {
  int8_t dosage[ORC_MAX_PLOIDY];
  haplotype_dosage_strided(dosage, labels, ploidy, 2, 2);
  int opt = 0;
  for (int h0 = 0; h0 < ploidy; h0++) {
    if (dosage[h0] == 0) continue;
    for (int h1 = h0 + 1; h1 < ploidy; h1++) {
      if (dosage[h1] == 0) continue;
      if (labels[h0 * 2] == labels[h1 * 2] || labels[h0 * 2 + 1] == labels[h1 * 2 + 1]) continue;
      int8_t *o = options + (size_t)opt * ploidy * 2;
      memcpy(o, labels, (size_t)ploidy * 2);
      o[h0 * 2] = labels[h1 * 2];
      o[h1 * 2] = labels[h0 * 2];
      opt++;
    }
  }
  return opt;
}

/* structural.py:181-237 */
int orc_dosage_step_n_options(const int8_t *labels, int ploidy) {
  int8_t hd[ORC_MAX_PLOIDY], sd[ORC_MAX_PLOIDY];
  haplotype_dosage_strided(hd, labels, ploidy, 2, 2);
  haplotype_dosage_strided(sd, labels, ploidy, 1, 2);
  int n = 0;
  for (int h0 = 0; h0 < ploidy; h0++) {
    if (hd[h0] == 0) continue;
    if (sd[h0] == 1) continue;
    for (int h1 = 0; h1 < ploidy; h1++) {
      if (sd[h1] == 0) continue;
      if (labels[h0 * 2] == labels[h1 * 2]) continue;
      n++;
    }
  }
  return n;
}

/* structural.py:240-307 */
int orc_dosage_step_options(const int8_t *labels, int ploidy, int8_t *options) {
  int8_t hd[ORC_MAX_PLOIDY], sd[ORC_MAX_PLOIDY];
  haplotype_dosage_strided(hd, labels, ploidy, 2, 2);
  haplotype_dosage_strided(sd, labels, ploidy, 1, 2);
  int opt = 0;
  for (int h0 = 0; h0 < ploidy; h0++) {
    if (hd[h0] == 0) continue;
    if (sd[h0] == 1) continue;
    for (int h1 = 0; h1 < ploidy; h1++) {
      if (sd[h1] == 0) continue;
      if (labels[h0 * 2] == labels[h1 * 2]) continue;
      int8_t *o = options + (size_t)opt * ploidy * 2;
      memcpy(o, labels, (size_t)ploidy * 2);
      o[h0 * 2] = labels[h1 * 2];
      opt++;
    }
  }
  return opt;
}

#define ORC_MAX_OPTIONS (ORC_MAX_PLOIDY * ORC_MAX_PLOIDY)

/* structural.py:433-576 up to (not including) random_choice. Returns n_options. */
static int interval_step_probs(orc_ctx *c, const int8_t *genotype, double llk, int start, int stop, int step_type,
                               double temp, double *probs, double *llks, int8_t *option_labels) {
  int ploidy = c->ploidy;
  int8_t labels[ORC_MAX_PLOIDY * 2];
  orc_haplotype_segment_labels(genotype, ploidy, c->n_pos, start, stop, labels);
  int n_options = (step_type == 0) ? orc_recombination_step_options(labels, ploidy, option_labels)
                                   : orc_dosage_step_options(labels, ploidy, option_labels);
  if (n_options == 0) return 0;
  double log_proposal_prob = log(1.0 / (double)n_options);
  double lprior = ctx_dosage_prior(c, genotype, c->n_pos);
  double log_accept[ORC_MAX_OPTIONS + 1];
  llks[n_options] = -INFINITY;
  log_accept[n_options] = -INFINITY;
  for (int i = 0; i < n_options; i++) {
    const int8_t *ol = option_labels + (size_t)i * ploidy * 2;
    int8_t hidx[ORC_MAX_PLOIDY];
    for (int h = 0; h < ploidy; h++) hidx[h] = ol[h * 2];
    if (c->stats) c->stats->structural_evals++;
    double llk_i = llk_structural_cached(c, genotype, hidx, start, stop);
    llks[i] = llk_i;
    double llk_ratio = llk_i - llk;
    double lprior_ratio = 0.0;
    if (!isnan(c->inbreeding)) lprior_ratio = ctx_dosage_prior(c, ol, 2) - lprior; /* structural.py:546 label rows */
    int n_return = (step_type == 0) ? orc_recombination_step_n_options(ol, ploidy) : orc_dosage_step_n_options(ol, ploidy);
    double log_return_prob = log(1.0 / (double)n_return);
    double lproposal_ratio = log_return_prob - log_proposal_prob;
    double mh = (llk_ratio + lprior_ratio) * temp + lproposal_ratio;
    log_accept[i] = fmin(0.0, mh);
  }
  double ln_opt = log((double)n_options);
  double sum = 0.0;
  for (int i = 0; i <= n_options; i++) {
    probs[i] = exp(log_accept[i] - ln_opt);
    sum += probs[i];
  }
  probs[n_options] = 1.0 - sum;
  return n_options;
}

int orc_interval_step_probabilities(const double *reads, int n_reads, int n_pos, int max_allele,
                                    const int8_t *genotype, int ploidy, double llk, int start, int stop, int step_type,
                                    double log_unique_haplotypes, double inbreeding, double temp,
                                    const int64_t *read_counts, double *probs, double *llks, int8_t *option_labels) {
  if (n_pos > ORC_MAX_POS || ploidy > ORC_MAX_PLOIDY) return ORC_ERR_LIMIT;
  orc_ctx c = {reads, n_reads, n_pos, max_allele, read_counts, NULL, ploidy, inbreeding, log_unique_haplotypes, NULL, NULL, NULL};
  return interval_step_probs(&c, genotype, llk, start, stop, step_type, temp, probs, llks, option_labels);
}

static double interval_step(orc_ctx *c, int8_t *genotype, double llk, int start, int stop, int step_type, double temp, int slot) {
  double probs[ORC_MAX_OPTIONS + 1], llks[ORC_MAX_OPTIONS + 1];
  int8_t option_labels[ORC_MAX_OPTIONS * ORC_MAX_PLOIDY * 2];
  int n_options = interval_step_probs(c, genotype, llk, start, stop, step_type, temp, probs, llks, option_labels);
  if (n_options == 0) return llk; /* structural.py:504-506: no RNG consumed */
  ORC_MLOG(c, genotype, 3, start, stop, step_type);
  int choice = choose_from(probs, n_options + 1, rng_double(c->rng, slot));
  if (choice < n_options) {
    int8_t hidx[ORC_MAX_PLOIDY];
    for (int h = 0; h < c->ploidy; h++) hidx[h] = option_labels[((size_t)choice * c->ploidy + h) * 2];
    orc_structural_change(genotype, c->ploidy, c->n_pos, hidx, start, stop);
    llk = llks[choice];
    ORC_MLOG(c, genotype, 2, 0, 0, 0);
  }
  return llk;
}

/* structural.py:22-71; returns number of intervals or <0 */
static int random_breaks(orc_rng *g, int slot, int breaks, int n, int (*intervals)[2]) {
  if (breaks >= n) return ORC_ERR_BREAKS;
  uint8_t ind[ORC_MAX_POS + 2];
  for (int i = 0; i <= n; i++) ind[i] = 1;
  ind[0] = 0;
  ind[n] = 0;
  for (int b = 0; b < breaks; b++) {
    int options[ORC_MAX_POS + 2], no = 0;
    for (int i = 0; i <= n; i++) if (ind[i]) options[no++] = i;
    if (no == 0) break;
    int point = options[rng_interval(g, slot, (uint32_t)(no - 1))]; /* np.random.choice(options) */
    ind[point] = 0;
  }
  int points[ORC_MAX_POS + 2], np_ = 0;
  for (int i = 0; i <= n; i++) if (!ind[i]) points[np_++] = i;
  for (int i = 0; i < breaks + 1; i++) {
    intervals[i][0] = points[i];
    intervals[i][1] = points[i + 1];
  }
  return breaks + 1;
}

/* structural.py:590-673 */
static double structural_compound_step(orc_ctx *c, int8_t *genotype, double llk, int (*intervals)[2], int n_intervals,
                                       int step_type, double temp, int slot) {
  int order[ORC_MAX_POS + 1];
  for (int i = 0; i < n_intervals; i++) order[i] = i;
  /* np.random.permutation(np.arange(n)): copy + Fisher-Yates */
  for (int i = n_intervals - 1; i >= 1; i--) {
    int k = (int)rng_interval(c->rng, slot, (uint32_t)i);
    int t = order[i]; order[i] = order[k]; order[k] = t;
  }
  for (int i = 0; i < n_intervals; i++) {
    int *iv = intervals[order[i]];
    llk = interval_step(c, genotype, llk, iv[0], iv[1], step_type, temp, slot);
  }
  return llk;
}

/* ------------------------------------------------------------------------- */
/* tempering (assemble/tempering.py)                                          */
/* ------------------------------------------------------------------------- */

/* tempering.py:10-58 */
static double chain_swap_acceptance(double llk_i, double lp_i, double temp_i, double llk_j, double lp_j, double temp_j) {
  double ui = llk_i + lp_i, uj = llk_j + lp_j;
  double f1 = (uj - ui) * temp_i;
  double f2 = (ui - uj) * temp_j;
  double a = exp(f1 + f2);
  if (a > 1.0) a = 1.0;
  return a;
}

/* tempering.py:61-151; i = cooler (current temp), j = warmer (previous temp) */
static void chain_swap_step(orc_ctx *c, int8_t *gi, double *llk_i, double temp_i, int8_t *gj, double *llk_j, double temp_j, int slot) {
  double prior_i = ctx_dosage_prior(c, gi, c->n_pos);
  double prior_j = ctx_dosage_prior(c, gj, c->n_pos);
  double acc = chain_swap_acceptance(*llk_i, prior_i, temp_i, *llk_j, prior_j, temp_j);
  double val = rng_double(c->rng, slot);
  if (acc >= val) {
    int8_t tmp[ORC_MAX_PLOIDY * ORC_MAX_POS];
    size_t n = (size_t)c->ploidy * c->n_pos;
    memcpy(tmp, gi, n);
    memcpy(gi, gj, n);
    memcpy(gj, tmp, n);
    double t = *llk_i; *llk_i = *llk_j; *llk_j = t;
  }
}

/* ------------------------------------------------------------------------- */
/* pre-sampling: snp_posterior, homozygosity, mean read dist                   */
/* ------------------------------------------------------------------------- */

/* snpcalling.py:14-70.  read_probs is a strided view reads[:, i, :] */
static int snp_posterior_strided(const double *read_probs, size_t rstride, int n_reads, int max_allele, int n_alleles,
                                 int ploidy, double inbreeding, const int64_t *read_counts, double *probs) {
  int64_t u_gens = orc_comb_with_replacement(n_alleles, ploidy);
  int64_t genotype[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) genotype[i] = 0;
  double *lp = (double *)malloc(sizeof(double) * (size_t)(u_gens > 0 ? u_gens : 1));
  for (int64_t i = 0; i < u_gens; i++) {
    double lprior = 0.0;
    if (!isnan(inbreeding)) lprior = orc_calling_log_genotype_prior(genotype, ploidy, n_alleles, inbreeding, NULL);
    /* log_likelihood on the 1-position view (snpcalling.py:61-65) */
    double llk = 0.0;
    if (n_reads == 0) {
      /* one all-NaN read: ln(sum_h 1/ploidy) */
      double rp = 0.0;
      for (int h = 0; h < ploidy; h++) rp += 1.0 / (double)ploidy;
      llk = log(rp);
    }
    for (int r = 0; r < n_reads; r++) {
      double rp = 0.0;
      for (int h = 0; h < ploidy; h++) {
        double prod = 1.0;
        double val = read_probs[(size_t)r * rstride + genotype[h]];
        if (!isnan(val)) prod *= val;
        rp += prod / (double)ploidy;
      }
      double l = log(rp);
      if (read_counts) l *= (double)read_counts[r];
      llk += l;
    }
    lp[i] = lprior + llk;
    orc_increment_genotype(genotype, ploidy);
  }
  normalise_log_probs(lp, (int)u_gens, probs);
  free(lp);
  (void)max_allele;
  return (int)u_gens;
}

int orc_snp_posterior(const double *read_probs, int n_reads, int max_allele, int n_alleles, int ploidy,
                      double inbreeding, const int64_t *read_counts, double *probs) {
  return snp_posterior_strided(read_probs, (size_t)max_allele, n_reads, max_allele, n_alleles, ploidy, inbreeding, read_counts, probs);
}

/* assemble/mcmc.py:494-541 */
void orc_homozygosity_probabilities(const double *reads, int n_reads, int n_pos, int max_allele, const int8_t *n_alleles,
                                    int ploidy, double inbreeding, const int64_t *read_counts, double *out) {
  for (int i = 0; i < n_pos * max_allele; i++) out[i] = 0.0;
  for (int i = 0; i < n_pos; i++) {
    int n = n_alleles[i];
    int64_t u = orc_comb_with_replacement(n, ploidy);
    double *probs = (double *)malloc(sizeof(double) * (size_t)(u > 0 ? u : 1));
    snp_posterior_strided(reads + (size_t)i * max_allele, (size_t)n_pos * max_allele, n_reads, max_allele, n, ploidy,
                          inbreeding, read_counts, probs);
    int64_t g[ORC_MAX_PLOIDY];
    for (int a = 0; a < n; a++) {
      for (int k = 0; k < ploidy; k++) g[k] = a;
      out[(size_t)i * max_allele + a] = probs[orc_genotype_alleles_as_index(g, ploidy)];
    }
    free(probs);
  }
}

/* numpy add.reduce over a short contiguous axis: first element, then the pairwise
 * (here: sequential, n<8) sum of the rest.  Used where the reference calls numpy (not numba). */
static double numpy_sum_small(const double *a, int n) {
  if (n == 0) return 0.0;
  double rest = 0.0;
  for (int i = 1; i < n; i++) rest += a[i];
  return n > 1 ? a[0] + rest : a[0];
}

/* assemble/mcmc.py:455-491 (pure numpy in the reference) */
void orc_read_mean_dist(const double *reads, int n_reads, int n_pos, int max_allele, double *dist) {
  for (int j = 0; j < n_pos; j++) {
    int n_nonzero = 0;
    uint8_t gap[ORC_MAX_ALLELE];
    for (int a = 0; a < max_allele; a++) {
      int all_nan = 1, all_zero = 1, cnt = 0;
      double tot = 0.0;
      for (int r = 0; r < n_reads; r++) {
        double v = reads[((size_t)r * n_pos + j) * max_allele + a];
        if (!isnan(v)) all_nan = 0;
      }
      for (int r = 0; r < n_reads; r++) {
        double v = reads[((size_t)r * n_pos + j) * max_allele + a];
        if (all_nan) v = 1.0;             /* mcmc.py:480 */
        if (!isnan(v)) { tot += v; cnt++; } /* nanmean: sequential over reads */
        if (!(v == 0.0)) all_zero = 0;    /* mcmc.py:484 (NaN != 0) */
      }
      gap[a] = (uint8_t)all_nan;
      dist[(size_t)j * max_allele + a] = tot / (double)cnt;
      if (!all_zero) n_nonzero++;
    }
    for (int a = 0; a < max_allele; a++)
      if (gap[a]) dist[(size_t)j * max_allele + a] = 1.0 / (double)n_nonzero;
    double s = numpy_sum_small(dist + (size_t)j * max_allele, max_allele);
    for (int a = 0; a < max_allele; a++) dist[(size_t)j * max_allele + a] /= s;
  }
}

/* jitutils.py:464-498: one haplotype drawn from dist [n_pos][A] */
static void sample_snv_alleles(orc_rng *g, int slot, const double *dist, int n_pos, int max_allele, int8_t *out) {
  for (int j = 0; j < n_pos; j++) {
    double d[ORC_MAX_ALLELE], s = 0.0;
    for (int a = 0; a < max_allele; a++) s += dist[(size_t)j * max_allele + a];
    for (int a = 0; a < max_allele; a++) d[a] = dist[(size_t)j * max_allele + a] / s;
    int ch = choose_from(d, max_allele, rng_double(g, slot));
    if (ch >= max_allele) ch = max_allele - 1;
    out[j] = (int8_t)ch;
  }
}

/* ------------------------------------------------------------------------- */
/* the sampler (assemble/mcmc.py:268-426) and the operator (103-265)          */
/* ------------------------------------------------------------------------- */

static int denovo_assembler(const orc_denovo_cfg *cfg, orc_ctx *c, const int8_t *genotype0, const double *break_dist,
                            int n_break_dist, int8_t *trace /*[steps][K][n_pos]*/, double *llk_trace) {
  int K = c->ploidy, M = c->n_pos, T = cfg->n_temps;
  size_t gsz = (size_t)K * M;
  int8_t genotypes[ORC_MAX_TEMPS][ORC_MAX_PLOIDY * ORC_MAX_POS];
  double llks[ORC_MAX_TEMPS];
  /* mcmc.py:294 */
  double luh = 0.0;
  for (int j = 0; j < M; j++) luh += log((double)c->n_alleles[j]);
  c->log_unique_haplotypes = luh;
  double llk0 = orc_log_likelihood(c->reads, c->n_reads, M, c->max_allele, genotype0, K, c->read_counts);
  for (int t = 0; t < T; t++) { memcpy(genotypes[t], genotype0, gsz); llks[t] = llk0; }
  /* mcmc.py:306-312 */
  c->cache = NULL;
  if (cfg->llk_cache_threshold >= 0 && (int64_t)K * M * c->n_reads > cfg->llk_cache_threshold) {
    int maxa = 0;
    for (int j = 0; j < M; j++) if (c->n_alleles[j] > maxa) maxa = c->n_alleles[j];
    c->cache = cache_new(K * M, maxa);
  }
  int rc = ORC_OK;
  int intervals[ORC_MAX_POS + 1][2];
  for (int i = 0; i < cfg->steps && rc == ORC_OK; i++) {
    for (int t = 0; t < T; t++) {
      double llk = llks[t];
      int8_t *g = genotypes[t];
      double temp = cfg->temperatures[t];
      if (isnan(llk)) { rc = ORC_ERR_NAN_LLK; break; }
      /* Philox mode: the draws of MCMC step i are numbered from i * ORC_STEP_DRAWS in each temperature's stream,
         whatever earlier steps consumed (the HIP kernels evaluate the steps of a settled chain independently) */
      c->rng->n[t] = (uint64_t)i * ORC_STEP_DRAWS;
      ORC_MLOG(c, g, 0, 0, 0, 0);
      llk = mutation_compound_step(c, g, llk, temp, t);
      if (rng_double(c->rng, t) <= cfg->p_recomb) {
        int nb = choose_from(break_dist, n_break_dist, rng_double(c->rng, t));
        int ni = random_breaks(c->rng, t, nb, M, intervals);
        if (ni < 0) { rc = ni; break; }
        llk = structural_compound_step(c, g, llk, intervals, ni, 0, temp, t);
      }
      if (rng_double(c->rng, t) <= cfg->p_partial_dosage) {
        int nb = choose_from(break_dist, n_break_dist, rng_double(c->rng, t));
        int ni = random_breaks(c->rng, t, nb, M, intervals);
        if (ni < 0) { rc = ni; break; }
        llk = structural_compound_step(c, g, llk, intervals, ni, 1, temp, t);
      }
      if (rng_double(c->rng, t) <= cfg->p_dosage) {
        intervals[0][0] = 0;
        intervals[0][1] = M;
        llk = structural_compound_step(c, g, llk, intervals, 1, 1, temp, t);
      }
      if (t > 0) {
        double llk_prev = llks[t - 1];
        chain_swap_step(c, g, &llk, temp, genotypes[t - 1], &llk_prev, cfg->temperatures[t - 1], t);
        llks[t - 1] = llk_prev;
      }
      llks[t] = llk;
    }
    if (rc != ORC_OK) break;
    memcpy(trace + (size_t)i * gsz, genotypes[T - 1], gsz);
    llk_trace[i] = llks[T - 1];
  }
  cache_free(c->cache);
  c->cache = NULL;
  return rc;
}

/* one chain: assemble/mcmc.py:163-265 */
static int mcmc_chain(const orc_denovo_cfg *cfg, orc_rng *rng, const double *reads, int n_reads, int n_pos, int max_allele,
                      const int64_t *read_counts, const int8_t *n_alleles, const int8_t *initial,
                      int8_t *genotypes_out, double *llks_out, orc_stats *stats) {
  int K = cfg->ploidy, S = cfg->steps;
  double *hom = (double *)malloc(sizeof(double) * (size_t)n_pos * max_allele);
  orc_homozygosity_probabilities(reads, n_reads, n_pos, max_allele, n_alleles, K, cfg->inbreeding, read_counts, hom);
  int het[ORC_MAX_POS], n_het = 0;
  int8_t fixed_allele[ORC_MAX_POS];
  for (int j = 0; j < n_pos; j++) {
    int any = 0;
    fixed_allele[j] = 0;
    for (int a = 0; a < max_allele; a++)
      if (hom[(size_t)j * max_allele + a] >= cfg->fix_homozygous) { any = 1; fixed_allele[j] = (int8_t)a; }
    if (!any) het[n_het++] = j;
    else if (0) fixed_allele[j] = 0;
  }
  free(hom);
  if (n_het == 0) {
    /* mcmc.py:189-199 */
    for (int i = 0; i < S; i++) {
      for (int h = 0; h < K; h++) memcpy(genotypes_out + ((size_t)i * K + h) * n_pos, fixed_allele, (size_t)n_pos);
      llks_out[i] = NAN;
    }
    return ORC_OK;
  }
  /* reads_het = reads[:, heterozygous] */
  double *reads_het = (double *)malloc(sizeof(double) * (size_t)n_reads * n_het * max_allele);
  for (int r = 0; r < n_reads; r++)
    for (int jj = 0; jj < n_het; jj++)
      memcpy(reads_het + ((size_t)r * n_het + jj) * max_allele, reads + ((size_t)r * n_pos + het[jj]) * max_allele,
             sizeof(double) * (size_t)max_allele);
  int8_t na_het[ORC_MAX_POS];
  for (int jj = 0; jj < n_het; jj++) na_het[jj] = n_alleles[het[jj]];
  /* initial genotype: mcmc.py:202-208 */
  int8_t genotype[ORC_MAX_PLOIDY * ORC_MAX_POS];
  if (!initial) {
    double *dist = (double *)malloc(sizeof(double) * (size_t)n_het * max_allele);
    orc_read_mean_dist(reads_het, n_reads, n_het, max_allele, dist);
    for (int h = 0; h < K; h++) sample_snv_alleles(rng, ORC_MAX_TEMPS, dist, n_het, max_allele, genotype + (size_t)h * n_het);
    free(dist);
  } else {
    memcpy(genotype, initial, (size_t)K * n_het);
  }
  /* break distribution: mcmc.py:211-217 */
  double bd[ORC_MAX_POS + 1];
  int nbd;
  if (cfg->n_intervals == 0) {
    nbd = n_het;
    memcpy(bd, cfg->break_table + (size_t)n_het * n_pos, sizeof(double) * (size_t)n_het);
  } else {
    nbd = cfg->n_intervals;
    for (int i = 0; i < nbd; i++) bd[i] = 0.0;
    bd[nbd - 1] = 1.0;
  }
  orc_ctx c = {reads_het, n_reads, n_het, max_allele, read_counts, na_het, K, cfg->inbreeding, 0.0, NULL, rng, stats};
  int8_t *trace = (int8_t *)malloc((size_t)S * K * n_het);
  int rc = denovo_assembler(cfg, &c, genotype, bd, nbd, trace, llks_out);
  if (rc == ORC_OK) {
    /* mcmc.py:251-265 re-insert fixed columns */
    for (int i = 0; i < S; i++)
      for (int h = 0; h < K; h++) {
        int8_t *row = genotypes_out + ((size_t)i * K + h) * n_pos;
        memcpy(row, fixed_allele, (size_t)n_pos);
        for (int jj = 0; jj < n_het; jj++) row[het[jj]] = trace[((size_t)i * K + h) * n_het + jj];
      }
  }
  free(trace);
  free(reads_het);
  return rc;
}

/* DenovoMCMC.fit: assemble/mcmc.py:103-161 */
int orc_denovo_fit(const orc_denovo_cfg *cfg, const double *reads, int n_reads, int n_pos, int max_allele,
                   const int64_t *read_counts, const int8_t *n_alleles, const int8_t *initial,
                   int8_t *genotypes_out, double *llks_out, orc_stats *stats) {
  if (cfg->ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS || max_allele > ORC_MAX_ALLELE || cfg->n_temps > ORC_MAX_TEMPS ||
      cfg->n_temps < 1)
    return ORC_ERR_LIMIT;
  /* mcmc.py:224-226 */
  for (int t = 1; t < cfg->n_temps; t++) if (cfg->temperatures[t] < cfg->temperatures[t - 1]) return ORC_ERR_BAD_ARG;
  if (cfg->temperatures[0] < 0.0 || cfg->temperatures[cfg->n_temps - 1] != 1.0) return ORC_ERR_BAD_ARG;
  double *nan_read = NULL;
  if (n_reads == 0) {
    /* mcmc.py:132-137 */
    n_reads = 1;
    nan_read = (double *)malloc(sizeof(double) * (size_t)n_pos * max_allele);
    for (int i = 0; i < n_pos * max_allele; i++) nan_read[i] = NAN;
    reads = nan_read;
    read_counts = NULL;
  }
  mt_state mt;
  if (cfg->rng_kind == ORC_RNG_NUMPY_MT19937) mt_seed(&mt, (uint32_t)cfg->seed);
  int rc = ORC_OK;
  size_t gsz = (size_t)cfg->steps * cfg->ploidy * n_pos;
  /* number of het positions is the same for every chain (deterministic); initial is [chains][K][n_het] */
  for (int ch = 0; ch < cfg->chains && rc == ORC_OK; ch++) {
    orc_rng rng;
    memset(&rng, 0, sizeof(rng));
    rng.kind = cfg->rng_kind;
    rng.seed = cfg->seed;
    rng.stream_id = cfg->stream_id;
    rng.chain = (uint32_t)ch;
    rng.mt = &mt;
    const int8_t *init_ch = NULL;
    if (initial) {
      /* n_het needed for the stride: recompute cheaply */
      double *hom = (double *)malloc(sizeof(double) * (size_t)n_pos * max_allele);
      orc_homozygosity_probabilities(reads, n_reads, n_pos, max_allele, n_alleles, cfg->ploidy, cfg->inbreeding, read_counts, hom);
      int n_het = 0;
      for (int j = 0; j < n_pos; j++) {
        int any = 0;
        for (int a = 0; a < max_allele; a++) if (hom[(size_t)j * max_allele + a] >= cfg->fix_homozygous) any = 1;
        if (!any) n_het++;
      }
      free(hom);
      init_ch = initial + (size_t)ch * cfg->ploidy * n_het;
    }
    rc = mcmc_chain(cfg, &rng, reads, n_reads, n_pos, max_allele, read_counts, n_alleles, init_ch,
                    genotypes_out + (size_t)ch * gsz, llks_out + (size_t)ch * cfg->steps, stats);
  }
  free(nan_read);
  return rc;
}

int orc_denovo_fit_batch(const orc_denovo_cfg *cfg, int n_units, int n_threads,
                         const double *reads, int n_reads, int n_pos, int max_allele,
                         const int64_t *read_counts, const int8_t *n_alleles,
                         int8_t *genotypes_out, double *llks_out, orc_stats *stats) {
  int rc_all = ORC_OK;
  size_t rsz = (size_t)n_reads * n_pos * max_allele;
  size_t gsz = (size_t)cfg->chains * cfg->steps * cfg->ploidy * n_pos;
  size_t lsz = (size_t)cfg->chains * cfg->steps;
  orc_stats total = {0, 0, 0, 0};
#ifdef __GLIBC__
  /* keep the per-chain tries and traces on the per-thread heaps: with the default thresholds every chain's
     growing trie is mmap()ed / munmap()ed, and hundreds of threads then serialise on the process's mmap lock */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
#endif
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
  {
    /* per-thread scratch when the caller does not want the traces (CPU-baseline timing) */
    int8_t *gscratch = genotypes_out ? NULL : (int8_t *)malloc(gsz);
    double *lscratch = llks_out ? NULL : (double *)malloc(lsz * sizeof(double));
    orc_stats mine = {0, 0, 0, 0};
    int rc_mine = ORC_OK;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int u = 0; u < n_units; u++) {
      orc_denovo_cfg c = *cfg;
      c.stream_id = cfg->stream_id + (uint64_t)u;
      int rc = orc_denovo_fit(&c, reads + (size_t)u * rsz, n_reads, n_pos, max_allele,
                              read_counts ? read_counts + (size_t)u * n_reads : NULL, n_alleles, NULL,
                              genotypes_out ? genotypes_out + (size_t)u * gsz : gscratch,
                              llks_out ? llks_out + (size_t)u * lsz : lscratch, &mine);
      if (rc != ORC_OK) rc_mine = rc;
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      if (rc_mine != ORC_OK) rc_all = rc_mine;
      total.llk_evals += mine.llk_evals;
      total.llk_cache_hits += mine.llk_cache_hits;
      total.mutation_evals += mine.mutation_evals;
      total.structural_evals += mine.structural_evals;
    }
    free(gscratch);
    free(lscratch);
  }
  if (stats) *stats = total;
  return rc_all;
}

/* ------------------------------------------------------------------------- */
/* exact caller (calling/exact.py)                                            */
/* ------------------------------------------------------------------------- */

static double llk_of_alleles(const double *reads, int n_reads, int n_pos, int max_allele, int ploidy,
                             const int8_t *haplotypes, const int64_t *alleles, const int64_t *read_counts) {
  /* log_likelihood(reads, haplotypes[genotype]) */
  int8_t g[ORC_MAX_PLOIDY * ORC_MAX_POS];
  for (int h = 0; h < ploidy; h++) memcpy(g + (size_t)h * n_pos, haplotypes + (size_t)alleles[h] * n_pos, (size_t)n_pos);
  return orc_log_likelihood(reads, n_reads, n_pos, max_allele, g, ploidy, read_counts);
}

/* exact.py:252-292: float32 store (exact.py:254) */
int orc_genotype_likelihoods(const double *reads, int n_reads, int n_pos, int max_allele, int ploidy,
                             const int8_t *haplotypes, int n_haps, const int64_t *read_counts, float *out, double *out64) {
  if (ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS) return ORC_ERR_LIMIT;
  int64_t G = orc_comb_with_replacement(n_haps, ploidy);
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) g[i] = 0;
  for (int64_t i = 0; i < G; i++) {
    double l = llk_of_alleles(reads, n_reads, n_pos, max_allele, ploidy, haplotypes, g, read_counts);
    if (out) out[i] = (float)l;
    if (out64) out64[i] = l;
    orc_increment_genotype(g, ploidy);
  }
  return ORC_OK;
}

/* exact.py:295-329 with the float32 likelihood array genotype_likelihoods returns: the joint values are
 * stored into a float32 array (exact.py:317); the log-sum-exp of jitutils.py:7-74 and the final exp run in float64
 * (see the comment in the body). */
int orc_genotype_posteriors_f32(const float *llks, int64_t G, int ploidy, int n_alleles, int has_prior,
                                double inbreeding, const double *frequencies, double *out) {
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) g[i] = 0;
  float *joint = (float *)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1));
  for (int64_t i = 0; i < G; i++) {
    double lpr = has_prior ? orc_calling_log_genotype_prior(g, ploidy, n_alleles, inbreeding, frequencies) : 0.0;
    joint[i] = (float)((double)llks[i] + lpr);
    orc_increment_genotype(g, ploidy);
  }
  /* exact.py:320-329 + jitutils.py:7-74 as the COMPILED reference runs them: the joint values are stored in the
     likelihoods' float32, but numba types the running log-denominator float64 (add_log_prob's `return -np.inf` branch
     unifies its result to float64), so the sum and the final exp(joint - denominator) are float64.  The reference's
     golden VCF simple.output.mixed_depth.call-exact.frequencies.posteriors.skiprare.vcf pins this: a float32
     accumulator leaves 4.4e-6 of the probability mass unaccounted for (SQ 53 instead of the golden's 60). */
  double acc = (double)joint[0];
  for (int64_t i = 1; i < G; i++) acc = orc_add_log_prob(acc, (double)joint[i]);
  for (int64_t i = 0; i < G; i++) out[i] = exp((double)joint[i] - acc);
  free(joint);
  return ORC_OK;
}

int orc_genotype_posteriors_f64(const double *llks, int64_t G, int ploidy, int n_alleles, int has_prior,
                                double inbreeding, const double *frequencies, double *out) {
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) g[i] = 0;
  for (int64_t i = 0; i < G; i++) {
    double lpr = has_prior ? orc_calling_log_genotype_prior(g, ploidy, n_alleles, inbreeding, frequencies) : 0.0;
    out[i] = llks[i] + lpr;
    orc_increment_genotype(g, ploidy);
  }
  double acc = out[0];
  for (int64_t i = 1; i < G; i++) acc = orc_add_log_prob(acc, out[i]);
  for (int64_t i = 0; i < G; i++) out[i] = exp(out[i] - acc);
  return ORC_OK;
}

/* exact.py:332-369 */
void orc_posterior_allele_frequencies_f64(const double *post, int64_t G, int ploidy, int n_alleles,
                                          double *freqs, double *counts, double *occur) {
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) g[i] = 0;
  for (int a = 0; a < n_alleles; a++) { counts[a] = 0.0; occur[a] = 0.0; }
  for (int64_t i = 0; i < G; i++) {
    double p = post[i];
    for (int j = 0; j < ploidy; j++) {
      int64_t a = g[j];
      counts[a] += p;
      if (j == 0 || a != g[j - 1]) occur[a] += p;
    }
    orc_increment_genotype(g, ploidy);
  }
  for (int a = 0; a < n_alleles; a++) freqs[a] = counts[a] / (double)ploidy;
}
/* combinations_with_replacement(support, remainder) in lexicographic order, itertools semantics */
static int next_cwr(int *idx, int r, int n) {
  int i = r - 1;
  while (i >= 0 && idx[i] == n - 1) i--;
  if (i < 0) return 0;
  int v = idx[i] + 1;
  for (int k = i; k < r; k++) idx[k] = v;
  return 1;
}

static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

/* exact.py:156-249 (+17-61, 64-105, 108-153) */
int orc_posterior_mode(const double *reads, int n_reads, int n_pos, int max_allele, int ploidy,
                       const int8_t *haplotypes, int n_haps, const int64_t *read_counts,
                       int has_prior, double inbreeding, const double *frequencies,
                       int64_t *mode_alleles, double *mode_llk, double *mode_prob, double *support_prob,
                       double *freqs, double *occur) {
  if (ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS) return ORC_ERR_LIMIT;
  int64_t G = orc_comb_with_replacement(n_haps, ploidy);
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) g[i] = 0;
  int64_t mode_idx = 0;
  double m_llk = -INFINITY, m_lj = -INFINITY, total = -INFINITY;
  for (int64_t i = 0; i < G; i++) {
    double llk = llk_of_alleles(reads, n_reads, n_pos, max_allele, ploidy, haplotypes, g, read_counts);
    double lpr = has_prior ? orc_calling_log_genotype_prior(g, ploidy, n_haps, inbreeding, frequencies) : 0.0;
    double lj = llk + lpr;
    if (lj > m_lj) { mode_idx = i; m_llk = llk; m_lj = lj; }
    total = orc_add_log_prob(total, lj);
    orc_increment_genotype(g, ploidy);
  }
  orc_index_as_genotype_alleles(mode_idx, ploidy, mode_alleles);
  *mode_llk = m_llk;
  *mode_prob = exp(m_lj - total);
  if (support_prob) {
    /* exact.py:64-105 */
    int64_t support[ORC_MAX_PLOIDY];
    int ns = 0;
    for (int i = 0; i < ploidy; i++) {
      int seen = 0;
      for (int k = 0; k < ns; k++) if (support[k] == mode_alleles[i]) seen = 1;
      if (!seen) support[ns++] = mode_alleles[i];
    }
    qsort(support, (size_t)ns, sizeof(int64_t), cmp_i64); /* np.unique sorts */
    int rem = ploidy - ns;
    int idx[ORC_MAX_PLOIDY];
    for (int i = 0; i < rem; i++) idx[i] = 0;
    double slj = -INFINITY;
    int more = 1;
    while (more) {
      int64_t tmp[ORC_MAX_PLOIDY];
      for (int i = 0; i < ns; i++) tmp[i] = support[i];
      for (int i = 0; i < rem; i++) tmp[ns + i] = support[idx[i]];
      qsort(tmp, (size_t)ploidy, sizeof(int64_t), cmp_i64);
      double llk = llk_of_alleles(reads, n_reads, n_pos, max_allele, ploidy, haplotypes, tmp, read_counts);
      double lpr = has_prior ? orc_calling_log_genotype_prior(tmp, ploidy, n_haps, inbreeding, frequencies) : 0.0;
      slj = orc_add_log_prob(slj, llk + lpr);
      more = (rem > 0) ? next_cwr(idx, rem, ns) : 0;
    }
    *support_prob = exp(slj - total);
  }
  if (freqs || occur) {
    /* exact.py:108-153 */
    double *f = (double *)calloc((size_t)n_haps, sizeof(double));
    double *o = (double *)calloc((size_t)n_haps, sizeof(double));
    for (int i = 0; i < ploidy; i++) g[i] = 0;
    for (int64_t i = 0; i < G; i++) {
      double llk = llk_of_alleles(reads, n_reads, n_pos, max_allele, ploidy, haplotypes, g, read_counts);
      double lpr = has_prior ? orc_calling_log_genotype_prior(g, ploidy, n_haps, inbreeding, frequencies) : 0.0;
      double prob = exp(llk + lpr - total);
      for (int k = 0; k < ploidy; k++) {
        int64_t a = g[k];
        f[a] += prob;
        if (k == 0 || a != g[k - 1]) o[a] += prob;
      }
      orc_increment_genotype(g, ploidy);
    }
    for (int a = 0; a < n_haps; a++) {
      if (freqs) freqs[a] = f[a] / (double)ploidy;
      if (occur) occur[a] = o[a];
    }
    free(f);
    free(o);
  }
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* `mchap call`: sampler over known haplotypes (calling/mcmc.py)              */
/* ------------------------------------------------------------------------- */

/* calling/utils.py:36-58 */
static int count_allele(const int64_t *g, int ploidy, int64_t allele) {
  int n = 0;
  for (int i = 0; i < ploidy; i++) n += g[i] == allele;
  return n;
}

/* calling/prior.py:30-52 (flat) and 55-113 */
double orc_log_genotype_allele_prior(const int64_t *genotype, int ploidy, int variable_allele, int64_t unique_haplotypes,
                                     int has_prior, double inbreeding, const double *frequencies) {
  if (!has_prior) return log((double)count_allele(genotype, ploidy, genotype[variable_allele]));
  if (inbreeding == 0.0) {
    if (!frequencies) return log(1.0 / (double)unique_haplotypes);
    return log(frequencies[genotype[variable_allele]]);
  }
  const double constant_sum = (double)(ploidy - 1);
  const double constant_ibs = (double)(count_allele(genotype, ploidy, genotype[variable_allele]) - 1);
  const double scale = (1.0 - inbreeding) / inbreeding;
  double sum_alpha, variable_alpha;
  if (!frequencies) {
    const double alpha = (1.0 / (double)unique_haplotypes) * scale;
    sum_alpha = constant_sum + alpha * (double)unique_haplotypes;
    variable_alpha = alpha + constant_ibs;
  } else {
    double s = 0.0; /* alphas.sum(): sequential (numba) */
    for (int64_t i = 0; i < unique_haplotypes; i++) s += frequencies[i] * scale;
    sum_alpha = constant_sum + s;
    variable_alpha = frequencies[genotype[variable_allele]] * scale + constant_ibs;
  }
  const double left = lgamma(sum_alpha) - lgamma(1.0 + sum_alpha);
  const double right = lgamma(1.0 + variable_alpha) - lgamma(variable_alpha);
  return left + right;
}

/* dict of calling/likelihood.py:36-78: genotype index of the SORTED alleles -> llk */
typedef struct {
  int64_t *keys;
  double *vals;
  int cap, n;
} call_cache;
static void call_cache_init(call_cache *c) {
  c->cap = 256;
  c->n = 0;
  c->keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)c->cap);
  c->vals = (double *)malloc(sizeof(double) * (size_t)c->cap);
  for (int i = 0; i < c->cap; i++) c->keys[i] = -1;
}
static void call_cache_free(call_cache *c) {
  free(c->keys);
  free(c->vals);
}
static int call_cache_slot(const call_cache *c, int64_t key) {
  uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
  int i = (int)(h >> 40) & (c->cap - 1);
  while (c->keys[i] != -1 && c->keys[i] != key) i = (i + 1) & (c->cap - 1);
  return i;
}
static void call_cache_put(call_cache *c, int64_t key, double v) {
  if (2 * (c->n + 1) > c->cap) {
    call_cache old = *c;
    c->cap *= 2;
    c->keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)c->cap);
    c->vals = (double *)malloc(sizeof(double) * (size_t)c->cap);
    for (int i = 0; i < c->cap; i++) c->keys[i] = -1;
    for (int i = 0; i < old.cap; i++)
      if (old.keys[i] != -1) {
        int s = call_cache_slot(c, old.keys[i]);
        c->keys[s] = old.keys[i];
        c->vals[s] = old.vals[i];
      }
    free(old.keys);
    free(old.vals);
  }
  int s = call_cache_slot(c, key);
  if (c->keys[s] == -1) c->n++;
  c->keys[s] = key;
  c->vals[s] = v;
}

typedef struct {
  const double *reads;
  int n_reads, n_pos, max_allele;
  const int64_t *read_counts;
  const int8_t *haplotypes;
  int n_haps, ploidy;
  int has_prior;
  double inbreeding;
  const double *frequencies;
  call_cache *cache; /* nullable */
} call_ctx;

/* calling/likelihood.py:36-78 */
static double call_llk(const call_ctx *c, const int64_t *g) {
  if (!c->cache) return llk_of_alleles(c->reads, c->n_reads, c->n_pos, c->max_allele, c->ploidy, c->haplotypes, g, c->read_counts);
  int64_t sorted[ORC_MAX_PLOIDY];
  memcpy(sorted, g, sizeof(int64_t) * (size_t)c->ploidy);
  qsort(sorted, (size_t)c->ploidy, sizeof(int64_t), cmp_i64);
  const int64_t key = orc_genotype_alleles_as_index(sorted, c->ploidy);
  const int s = call_cache_slot(c->cache, key);
  if (c->cache->keys[s] == key) return c->cache->vals[s];
  const double v = llk_of_alleles(c->reads, c->n_reads, c->n_pos, c->max_allele, c->ploidy, c->haplotypes, g, c->read_counts);
  call_cache_put(c->cache, key, v);
  return v;
}

static double call_genotype_prior(const call_ctx *c, const int64_t *g) {
  if (!c->has_prior) return 0.0;
  return orc_calling_log_genotype_prior(g, c->ploidy, c->n_haps, c->inbreeding, c->frequencies);
}

/* calling/mcmc.py:143-229 */
static void gibbs_options(const call_ctx *c, int64_t *g, int k, double *llks, double *lpriors, double *probs) {
  const int64_t current = g[k];
  const int H = c->n_haps;
  double joint[256];
  for (int a = 0; a < H; a++) {
    g[k] = a;
    lpriors[a] = orc_log_genotype_allele_prior(g, c->ploidy, k, H, c->has_prior, c->inbreeding, c->frequencies);
    llks[a] = call_llk(c, g);
    joint[a] = llks[a] + lpriors[a];
  }
  normalise_log_probs(joint, H, probs);
  g[k] = current;
}

/* calling/mcmc.py:15-140 */
static void mh_options(const call_ctx *c, int64_t *g, int k, double *llks, double *lpriors, double *probs) {
  const int64_t current = g[k];
  const int H = c->n_haps;
  const int copies = count_allele(g, c->ploidy, g[k]);
  const double lprior = call_genotype_prior(c, g);
  const double llk = call_llk(c, g);
  double lprop[256];
  for (int a = 0; a < H; a++) {
    if (g[k] == a) { /* g[k] is reset to `current` only at the end: the comparison is against the running value */
      lprop[a] = 0.0;
      lpriors[a] = lprior;
      llks[a] = llk;
    } else {
      g[k] = a;
      lpriors[a] = call_genotype_prior(c, g);
      llks[a] = call_llk(c, g);
      lprop[a] = log((double)count_allele(g, c->ploidy, g[k]) / (double)copies);
    }
  }
  double sum = 0.0;
  for (int a = 0; a < H; a++) {
    const double r = (llks[a] - llk) + (lpriors[a] - lprior) + lprop[a];
    probs[a] = exp(r < 0.0 ? r : 0.0);
  }
  probs[current] = 0.0;
  for (int a = 0; a < H; a++) probs[a] /= (double)(H - 1);
  for (int a = 0; a < H; a++) sum += probs[a]; /* probabilities_array.sum() */
  probs[current] = 1.0 - sum;
  g[k] = current;
}

int orc_call_step_options(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                          const int8_t *haplotypes, int n_haps, const int64_t *genotype, int ploidy, int variable_allele,
                          int step_type, int has_prior, double inbreeding, const double *frequencies, double *llks,
                          double *lpriors, double *probs) {
  if (ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS || n_haps > 256) return ORC_ERR_LIMIT;
  call_ctx c = {reads, n_reads, n_pos, max_allele, read_counts, haplotypes, n_haps, ploidy, has_prior, inbreeding, frequencies, NULL};
  int64_t g[ORC_MAX_PLOIDY];
  memcpy(g, genotype, sizeof(int64_t) * (size_t)ploidy);
  if (step_type == 0) gibbs_options(&c, g, variable_allele, llks, lpriors, probs);
  else mh_options(&c, g, variable_allele, llks, lpriors, probs);
  return ORC_OK;
}

/* calling/mcmc.py:393-453 */
int orc_greedy_caller(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                      const int8_t *haplotypes, int n_haps, int ploidy, int has_prior, double inbreeding,
                      const double *frequencies, int64_t *genotype_out) {
  if (ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS) return ORC_ERR_LIMIT;
  int64_t g[ORC_MAX_PLOIDY];
  for (int i = 0; i < ploidy; i++) {
    const int k = i + 1;
    double best = -INFINITY;
    int64_t best_allele = -1;
    for (int a = 0; a < n_haps; a++) {
      g[i] = a;
      const double llk = llk_of_alleles(reads, n_reads, n_pos, max_allele, k, haplotypes, g, read_counts);
      const double lprior = has_prior ? orc_calling_log_genotype_prior(g, k, n_haps, inbreeding, frequencies) : 0.0;
      const double lprob = llk + lprior;
      if (lprob > best) {
        best = lprob;
        best_allele = a;
      }
    }
    g[i] = best_allele;
  }
  qsort(g, (size_t)ploidy, sizeof(int64_t), cmp_i64);
  memcpy(genotype_out, g, sizeof(int64_t) * (size_t)ploidy);
  return ORC_OK;
}

/* calling/mcmc.py:232-327 */
static double call_compound_step(const call_ctx *c, orc_rng *rng, int64_t *g, int step_type) {
  const int K = c->ploidy, H = c->n_haps;
  double llks[256], lpriors[256], probs[256];
  int order[ORC_MAX_PLOIDY];
  for (int i = 0; i < K; i++) order[i] = i;
  for (int i = K - 1; i >= 1; i--) { /* np.random.shuffle(order) */
    const int j = (int)rng_interval(rng, 0, (uint32_t)i);
    const int t = order[i];
    order[i] = order[j];
    order[j] = t;
  }
  int choice = 0;
  for (int j = 0; j < K; j++) {
    const int k = order[j];
    if (step_type == 0) gibbs_options(c, g, k, llks, lpriors, probs);
    else mh_options(c, g, k, llks, lpriors, probs);
    choice = choose_from(probs, H, rng_double(rng, 0));
    g[k] = choice;
  }
  qsort(g, (size_t)K, sizeof(int64_t), cmp_i64);
  return llks[choice < H ? choice : H - 1];
}

int orc_call_mcmc(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                  const int8_t *haplotypes, int n_haps, int ploidy, int has_prior, double inbreeding,
                  const double *frequencies, int steps, int chains, int step_type, const int64_t *initial, int rng_kind,
                  uint64_t seed, uint64_t stream_id, int64_t *genotypes_out, double *llks_out) {
  if (ploidy > ORC_MAX_PLOIDY || n_pos > ORC_MAX_POS || n_haps > 256) return ORC_ERR_LIMIT;
  if (step_type != 0 && step_type != 1) return ORC_ERR_BAD_ARG;
  int64_t init[ORC_MAX_PLOIDY];
  if (initial) memcpy(init, initial, sizeof(int64_t) * (size_t)ploidy);
  else orc_greedy_caller(reads, n_reads, n_pos, max_allele, read_counts, haplotypes, n_haps, ploidy, has_prior, inbreeding, frequencies, init);
  mt_state mt;
  if (rng_kind == ORC_RNG_NUMPY_MT19937) mt_seed(&mt, (uint32_t)seed);
  for (int ch = 0; ch < chains; ch++) {
    orc_rng rng;
    memset(&rng, 0, sizeof(rng));
    rng.kind = rng_kind;
    rng.seed = seed;
    rng.stream_id = stream_id;
    rng.chain = (uint32_t)ch;
    rng.mt = &mt;
    call_cache cache;
    call_cache_init(&cache);
    call_ctx c = {reads, n_reads, n_pos, max_allele, read_counts, haplotypes, n_haps, ploidy, has_prior, inbreeding, frequencies, &cache};
    int64_t g[ORC_MAX_PLOIDY];
    memcpy(g, init, sizeof(int64_t) * (size_t)ploidy);
    for (int s = 0; s < steps; s++) {
      const double llk = call_compound_step(&c, &rng, g, step_type);
      llks_out[(size_t)ch * steps + s] = llk;
      memcpy(genotypes_out + ((size_t)ch * steps + s) * ploidy, g, sizeof(int64_t) * (size_t)ploidy);
    }
    call_cache_free(&cache);
  }
  return ORC_OK;
}
