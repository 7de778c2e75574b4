/*
 * mchap_hip.h -- C ABI of libmchap_hip.so: MI355X (gfx950) kernels for MCHap's per-locus
 * MCMC haplotype assembler and exact genotype caller.
 *
 * The reference (PlantandFoodResearch/MCHap v0.11.1) has no FFI layer: its operator
 * boundary is Python (paths relative to /root/reference/mchap/):
 *   - DenovoMCMC.fit(reads, read_counts, initial) -> GenotypeMultiTrace   assemble/mcmc.py:24-161
 *   - calling.exact.genotype_likelihoods / genotype_posteriors /
 *     posterior_mode / posterior_allele_frequencies                      calling/exact.py:156-369
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows
 * the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions: plain pointers and sizes, caller allocates every output, the library never
 * retains a pointer.  Return value 0 on success, negative MCHAP_ERR_* otherwise (see
 * mchap_last_error()).  `*_device` entry points take DEVICE pointers and a hipStream_t
 * (passed as void*), enqueue work and return without synchronising; the plain entry points
 * take HOST pointers and are synchronous.
 */
#ifndef MCHAP_HIP_H
#define MCHAP_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCHAP_MAX_TEMPS 16
#define MCHAP_MAX_PLOIDY 8   /* the fast de novo samplers (nibble-packed labels) and the one-word forms of the device posterior summary; reference has no limit */
#define MCHAP_MAX_PLOIDY_DENOVO 15  /* ... the de novo sampler: ploidies 9 to 15 run on the general lanes-over-chains kernel (packs of
                                       sixteen nibbles; a dose of 16 does not fit one), with two trace words per haplotype */
#define MCHAP_MAX_POS 126    /* SNVs per unit.  Up to 62 SNVs and 64 bits of sampled alleles per haplotype (1 bit per biallelic, 2 per
                                tri- / tetra-allelic, 3 beyond) the fast samplers run; wider units -- up to 126 SNVs and 128 bits -- run on
                                the lanes-over-chains sampler with 128-bit haplotype words and their traces hold TWO uint64 words per
                                haplotype (mchap_denovo_trace_words_per_haplotype).  The reference has no limit. */
#define MCHAP_MAX_ALLELE 8
#define MCHAP_MAX_READS 4096 /* rows per unit after de-duplication (kernels 3 and 5; kernels 1 and 2: 1024) */

enum {
  MCHAP_OK = 0,
  MCHAP_ERR_NAN_LLK = -1,   /* ValueError("Encountered log likelihood of nan"), assemble/mcmc.py:330-331 */
  MCHAP_ERR_BAD_ARG = -2,   /* AssertionError family: shapes, temperatures (assemble/mcmc.py:134,207,220,225-226) */
  MCHAP_ERR_BREAKS = -3,    /* ValueError("breaks must be smaller then n"), assemble/structural.py:49-50 */
  MCHAP_ERR_LIMIT = -4,     /* shape exceeds what the kernels support (ploidy, packed haplotype width, LDS) */
  MCHAP_ERR_HIP = -5,       /* HIP runtime failure */
  MCHAP_ERR_NO_DEVICE = -6
};

/* per-unit status written by the sampler kernel */
enum {
  MCHAP_UNIT_OK = 0,
  MCHAP_UNIT_ALL_FIXED = 1, /* every position fixed homozygous: constant trace, llk = NaN (assemble/mcmc.py:189-199) */
  MCHAP_UNIT_NAN_LLK = 2,
  MCHAP_UNIT_BREAKS = 3,
  MCHAP_UNIT_BAD_INITIAL = 4 /* initial.shape != (ploidy, n_het_base): AssertionError, assemble/mcmc.py:207 */
};

/* Measurement / test knobs of the sampler (optional: mchap_denovo_cfg.tuning == NULL selects every default).  Results never
 * depend on them -- every setting produces the same traces (tests/test_gpu_denovo.py runs the extremes) -- only the time does.
 * A zero field means "default". */
typedef struct mchap_denovo_tuning {
  int32_t cache_slots;   /* {tag, llk} entries per chain of the likelihood cache: a power of two in 64..65536 (default: what 4 GiB
                            over the batch's chains allow, 1024..65536) */
  int32_t flags;         /* 1: no mutation memo, 2: no interval memo, 4: no coded read table, 8: no product reuse,
                            16: no LDS base-product cache, 32: skip the phased sampler's table completion (timing only: with
                            pipe_stop), 64 (libmchap_hip_test.so only): the tables completed by denovo_fill_kernel, one lane per request,
                            128: units without information (all gaps) are evaluated like any other,
                            256: never the instantiation that evaluates the requests of shallow units (<= 64 reads) side by side,
                            512: no haplotype-product rows in the workspace for deep units (> 256 reads),
                            1024: table completion inside the exporting launch, 8192: no LDS front cache of a chain's likelihoods,
                            16384: a ladder's replicas on one wavefront, 32768: no completion memo across chunks, 65536: caches
                            cleared per call instead of epoch tags, 131072: one wavefront per chain in every coasting launch,
                            262144: a round's misses behind a sub-step known to move are evaluated as well, 524288: no decision
                            contexts per genotype for the phased sampler's resumed chains, 1048576: a logarithm per read even where
                            the read weights are 0 / 1 (else one per group of four reads: the log likelihoods then differ in the last
                            bits, the traces do not)
                            (DESIGN.md section 4 names each) */
  int32_t spec_group;    /* kernel 3: lanes per chain, 16 / 32 / 64 (default: the smallest the shape allows) */
  int32_t pipe_first;    /* kernel 5: MCMC steps before the first hand-over (default ploidy * n_pos / 10, clamped to 3..32) */
  int32_t pipe_resume;   /*           steps a handed-back chain runs before it is handed over again (default 8) */
  int32_t pipe_rounds;   /*           resume rounds + 1 before the rest runs to the end in kernel 3 (default 2 + 1; 1 = none) */
  int32_t pipe_max;      /*           cap of a launch's extension while a chain of the wave is unsettled (default 64) */
  int32_t pipe_parts;    /*           wavefronts per chain completing tables when the chains are few (default 8) */
  int32_t prep_lds_limit; /* bytes of table the prepare pass copies to LDS (default 8192) */
  int32_t pipe_stop;     /* n > 0: stop after the n-th coasting launch (traces incomplete: timing / inspection of the phases) */
  int32_t reserved[6];
} mchap_denovo_tuning;

/* The fields of the DenovoMCMC dataclass (assemble/mcmc.py:24-40) that are common to a batch. */
typedef struct mchap_denovo_cfg {
  int32_t steps;                            /* steps */
  int32_t chains;                           /* chains */
  int32_t n_temps;                          /* len(temperatures) */
  int32_t n_intervals;                      /* n_intervals, 0 == None */
  double temperatures[MCHAP_MAX_TEMPS];     /* ascending, last == 1.0 (assemble/mcmc.py:224-226) */
  double fix_homozygous;                    /* fix_homozygous */
  double p_recomb;                          /* recombination_step_probability */
  double p_partial_dosage;                  /* partial_dosage_step_probability */
  double p_dosage;                          /* dosage_step_probability */
  uint64_t seed;                            /* random_seed: Philox4x32-10 key */
  const double *break_table;                /* HOST pointer, [(max_pos+1) x max_pos]: row m = Beta(alpha,beta) CDF
                                               increments over m het bases (assemble/mcmc.py:429-452) */
  int32_t max_pos;                          /* leading dimension of break_table */
  int32_t llk_cache;                        /* llk_cache_threshold >= 0: 1 = cache likelihoods per chain, 0 = recompute */
  int32_t kernel;                           /* 0 = default (5 when the batch's shape allows, else 3, else 2), 2 = lanes over
                                               chains, 3 = speculative sub-steps with a group of lanes per chain, 5 = phased:
                                               kernel 3 for a chain's first steps, then its settled stretches one lane per
                                               MCMC step (single temperature, one ploidy per launch).  Identical results
                                               whichever runs.  (1 = wavefront per chain and 4 = settle/steady pipeline are
                                               earlier designs kept for the parity suite: libmchap_hip_test.so only.) */
  int32_t cache_epoch;                      /* 1 .. 2^31 - 2: a number no other fit on memory this fit's workspace may reuse has carried
                                               (the likelihood caches of packed genotypes are tagged with it instead of being
                                               cleared: what an earlier call left in the workspace never matches).  The Python
                                               host hands out ONE process-wide sequence to every library copy it has loaded
                                               (mchap_amd/_lib.py next_cache_epoch).  0: the library's own counter -- unique among
                                               the fits of one loaded copy of the library, which is what a C caller has */
  const mchap_denovo_tuning *tuning;        /* HOST pointer or NULL */
  void *timer;                              /* mchap_timer_create() handle or NULL: the sampler launches of this call are
                                               bracketed by HIP events on the call's stream (not the prepare pass) */
} mchap_denovo_cfg;

/* One unit = one (locus x sample) call of DenovoMCMC.fit.  Offsets are in ELEMENTS of the
 * corresponding batch buffer. */
typedef struct mchap_unit {
  int64_t reads_off;     /* into reads: this unit's float64 [n_reads][n_pos][max_allele] tensor, C order */
  int64_t counts_off;    /* into read_counts, or -1 == None */
  int64_t nalleles_off;  /* into n_alleles: int8 [n_pos] */
  int64_t initial_off;   /* into initial: int8 [chains][ploidy][n_het], or -1 == None */
  int64_t trace_off;     /* into trace_words: uint64 [chains][steps][ploidy] */
  int64_t llk_off;       /* into llks: float64 [chains][steps] */
  int64_t fixed_off;     /* into fixed_alleles: int8 [n_pos] */
  int32_t n_reads;
  int32_t n_pos;
  int32_t max_allele;
  int32_t ploidy;
  double inbreeding;     /* NaN == None (flat prior) */
  uint64_t stream_id;    /* RNG stream of the unit: results do not depend on batch order or sharding */
  int32_t initial_n_het; /* last dimension of the caller's `initial` array; the unit ends with MCHAP_UNIT_BAD_INITIAL
                            (nothing sampled, `initial` not read) unless it equals the number of non-fixed positions */
  int32_t reserved;
} mchap_unit;

/* Replaces DenovoMCMC.fit (assemble/mcmc.py:103-161) for a batch of units.
 *
 * Outputs per unit:
 *   trace_words  uint64 [chains][steps][ploidy]: the cold-chain genotype of every step with its haplotypes
 *                packed over the NON-fixed positions (position 0 most significant, `bits` bits per allele,
 *                bits = 1/2/3 for max_allele <= 2/4/8) and sorted ascending -- i.e. already in the
 *                canonical order GenotypeMultiTrace.__post_init__ produces (assemble/classes.py:265-278).
 *                A batch with a unit wider than the fast samplers take (MCHAP_MAX_POS above) holds W = 2 words per haplotype,
 *                uint64 [chains][steps][ploidy][2], most significant word first (W for a batch:
 *                mchap_denovo_trace_words_per_haplotype; the caller sizes trace_words and its trace_off by it).
 *   llks         float64 [chains][steps]  (assemble/mcmc.py:418-425)
 *   fixed_alleles int8 [n_pos]: -1 for sampled positions, else the allele fixed as homozygous
 *                (assemble/mcmc.py:168-182,251-265); together with trace_words this is the full trace.
 *   status       int32 per unit, MCHAP_UNIT_*.
 * All pointers are device pointers; `units` too.  `stream` is a hipStream_t. */
int mchap_denovo_fit_batch_device(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                  const mchap_unit *units_host, const double *reads, const int64_t *read_counts,
                                  const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words,
                                  double *llks, int8_t *fixed_alleles, int32_t *status, void *workspace,
                                  int64_t workspace_bytes, void *stream);

/* The same with the compact input the reference's encoders start from (encoding/integer/transcode.py:16-77,
 * io/bam.py:251-289): `calls` int8 [n_reads][n_pos] per unit (allele index, < 0 = gap; mchap_unit.reads_off counts
 * ELEMENTS of this array), `quals` int16 of the same shape or NULL, and `qual_prob[q]` = probability that a call with
 * base quality q is correct, i.e. (1 - error_rate) * (1 - 10^(-q/10)), computed by the caller exactly as the
 * reference does (qual_prob[0] = 1 - error_rate is used for every call when quals == NULL; qualities beyond the
 * table use its last entry).  The probability tensor is formed on the device by the prepare pass: the called
 * allele gets p, the others (1 - p) / 3, a gap NaN, alleles >= n_alleles[j] zero.  5x fewer input bytes than the
 * float64 tensor; same traces.  Every shipped sampler takes it (kernels 2, 3 and the phased sampler 5: they share the prepare
 * pass); kernel 1 of the parity suite does not. */
int mchap_denovo_fit_batch_calls_device(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_dev,
                                        const mchap_unit *units_host, const int8_t *calls, const int16_t *quals,
                                        const double *qual_prob, int qual_prob_len, const int64_t *read_counts,
                                        const int8_t *n_alleles, const int8_t *initial, uint64_t *trace_words,
                                        double *llks, int8_t *fixed_alleles, int32_t *status, void *workspace,
                                        int64_t workspace_bytes, void *stream);

/* Bytes of device workspace the sampler needs: per-chain likelihood caches (the counterpart of the reference's llk
 * cache, assemble/likelihood.py:151-305; results-neutral) and, for kernel 0/2, the transposed read tensors and
 * per-unit tables written by its prepare pass. */
int64_t mchap_denovo_workspace_bytes(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host);

/* Same with host pointers: allocates device buffers, copies, runs, synchronises, copies back. */
int mchap_denovo_fit_batch(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units,
                           const double *reads, int64_t reads_len, const int64_t *read_counts, int64_t counts_len,
                           const int8_t *n_alleles, int64_t nalleles_len, const int8_t *initial, int64_t initial_len,
                           uint64_t *trace_words, int64_t trace_len, double *llks, int64_t llks_len,
                           int8_t *fixed_alleles, int64_t fixed_len, int32_t *status);

/* Test hook: log_likelihood (assemble/likelihood.py:17-70) of n_genotypes genotypes of one unit.
 * Host pointers. genotypes int8 [n_genotypes][ploidy][n_pos]. */
int mchap_log_likelihood_batch(const double *reads, int n_reads, int n_pos, int max_allele,
                               const int64_t *read_counts, const int8_t *genotypes, int n_genotypes, int ploidy,
                               double *llks_out);

/* Test hook: the logarithm the likelihood kernels take of their per-read terms (csrc/read_log.hpp: < 1 ulp; 0 -> -inf,
 * negative or NaN -> NaN), for n arguments.  Host pointers. */
int mchap_read_log_batch(const double *x, int64_t n, double *out);
/* Test hook: the logarithm of the product of `group` (1..4) per-read terms as ONE logarithm -- mantissas multiplied, exponents
 * summed (csrc/read_log.hpp read_log_product) -- which is what the likelihood kernels take of a lane's reads where the read weights
 * are 0 / 1: out[i] for x[i * group .. i * group + group - 1].  Host pointers. */
int mchap_read_log_product_batch(const double *x, int64_t n, int group, double *out);
/* Test hook: the 64-lane sum every likelihood goes through (csrc/denovo_kernel.hpp wave_sum: the XOR butterfly's tree, its last
 * four steps through DPP row rotations): out[w] = sum of x[64 w .. 64 w + 63].  Host pointers. */
int mchap_wave_sum_batch(const double *x, int64_t n_waves, double *out);

/* Posterior summary of a batch of traces: replaces GenotypeMultiTrace.burn(n).posterior() and the
 * mode/support statistics the assemble program reads from it (assemble/classes.py:280-325,87-128,194-205;
 * application/assemble.py:144-157).  Device pointers.
 *   post_words  uint64 [n_units][max_states][ploidy_max]  distinct genotypes, probability descending
 *   post_counts int32  [n_units][max_states]              occurrences after burn-in over all chains
 *   post_n      int32  [n_units]                          number of distinct genotypes (may exceed max_states)
 *   mode_stats  float64 [n_units][2]                      (support probability SPM, mode genotype probability GPM)
 *   mode_index  int32  [n_units]                          rank (row of post_words when < max_states) of the mode genotype
 *                                                         of the mode support
 *   mode_words  uint64 [n_units][ploidy_max] or NULL      that genotype itself, whatever its rank
 *   mode_count  int32  [n_units] or NULL                  its occurrences
 * post_n < 0: more than 512 distinct genotypes (-n: counts beyond the 512th are lost; fall back to the trace);
 * INT32_MIN: the unit's ploidy exceeds ploidy_max. */
int mchap_trace_posterior_batch_device(int n_units, const mchap_unit *units_dev, int steps, int chains, int burn,
                                       const uint64_t *trace_words, int max_states, int ploidy_max,
                                       uint64_t *post_words, int32_t *post_counts, int32_t *post_n,
                                       double *mode_stats, int32_t *mode_index, uint64_t *mode_words,
                                       int32_t *mode_count, void *stream);

/* The same summary for the few units of a batch whose chains visited more distinct genotypes than the batch call keeps
 * (post_n < 0: shallow or empty samples whose chains wander), with a table of `cap` states per unit instead of 512 --
 * up to mchap_trace_posterior_max_states(ploidy_max), what 160 KB of LDS hold; cap >= chains * (steps - burn) can never
 * overflow.  unit_list int32 [n_list] (device): the units; post_words uint64 [n_list][cap][ploidy_max] and post_counts
 * int32 [n_list][cap] are indexed by list position, the per-unit outputs (post_n, mode_*) by unit, overwriting the batch
 * call's values.  So the program never has to download a trace (assemble/classes.py:280-325 on the host). */
int mchap_trace_posterior_max_states(int ploidy_max);
int mchap_trace_posterior_listed_device(int n_list, const int32_t *unit_list_dev, const mchap_unit *units_dev, int steps, int chains,
                                        int burn, const uint64_t *trace_words, int cap, int ploidy_max, uint64_t *post_words,
                                        int32_t *post_counts, int32_t *post_n, double *mode_stats, int32_t *mode_index,
                                        uint64_t *mode_words, int32_t *mode_count, void *stream);

/* `mchap call`: the sampler's traces summarised on the device (round 5).  The traces of mchap_call_mcmc_batch_device -- int64
 * genotypes [U][chains][steps][ploidy], alleles ascending -- are states of `ploidy` words, so GenotypeAllelesMultiTrace.burn(n)
 * .posterior() and PosteriorGenotypeAllelesDistribution.mode(genotype_support=True) (calling/classes.py:180-196, 303-362) are
 * mchap_trace_posterior_batch_device / _listed_device above as they stand (units: ploidy and trace_off = u * chains * steps *
 * ploidy; post_words are allele indices), ploidy <= MCHAP_MAX_PLOIDY.  The calling classes' replicate_incongruence
 * (calling/classes.py:231-263) differs from the assembler's: chains whose mode support reaches the threshold are compared by their
 * mode GENOTYPES, and the code is 2 when those genotypes together hold more distinct alleles than the ploidy.  mci int32 [n_units]:
 * 0 / 1 / 2, -1 if a chain visited more distinct genotypes than the table holds (then the listed form with a larger table). */
int mchap_call_incongruence_batch_device(int n_units, const mchap_unit *units_dev, int steps, int chains, int burn,
                                         const int64_t *genotypes, int ploidy_max, double threshold, int32_t *mci, void *stream);
int mchap_call_incongruence_listed_device(int n_list, const int32_t *unit_list_dev, const mchap_unit *units_dev, int steps, int chains,
                                          int burn, const int64_t *genotypes, int cap, int ploidy_max, double threshold, int32_t *mci,
                                          void *stream);

/* Exact caller: replaces calling.exact.genotype_likelihoods (calling/exact.py:266-292): llks_out float32 [G] as the
 * reference stores them and / or llks64_out float64 [G] (the unrounded values), G = C(n_haps + ploidy - 1, ploidy) genotypes in
 * VCF order; either may be NULL.  Host pointers, one unit. */
int mchap_exact_genotype_likelihoods(const double *reads, int n_reads, int n_pos, int max_allele,
                                     const int64_t *read_counts, const int8_t *haplotypes, int n_haps, int ploidy,
                                     float *llks_out, double *llks64_out);

/* Replaces calling.exact.genotype_posteriors (calling/exact.py:295-329) on a stored likelihood array (float32 as
 * genotype_likelihoods returns it, or float64): priors in VCF order, normalisation; float64 result array whose
 * values carry float32 precision when the input is float32, as in the reference.  Host pointers. */
int mchap_exact_genotype_posteriors(const void *llks, int is_f32, int64_t n_genotypes, int ploidy, int n_alleles,
                                    int has_prior, double inbreeding, const double *frequencies, double *post_out);

/* Replicate incongruence of every unit's chains (the MCI field of `mchap assemble`): replaces
 * GenotypeMultiTrace.replicate_incongruence(threshold) (assemble/classes.py:341-376) on the traces written by
 * mchap_denovo_fit_batch_device.  mci[u] = 0 none, 1 incongruence, 2 incongruence whose union of haplotypes is larger
 * than the first qualifying chain's support (putative CNV; the reference's `len(alleles[0])`, classes.py:371-375), -1 if a chain visited more distinct genotypes than the kernel keeps (512).  Device pointers. */
int mchap_trace_incongruence_batch_device(int n_units, const mchap_unit *units_dev, int steps, int chains, int burn,
                                          const uint64_t *trace_words, int ploidy_max, double threshold, int32_t *mci,
                                          void *stream);

/* ... and for a list of units with a table of `cap` states per chain (mci[u] == -1 after the batch call), as
 * mchap_trace_posterior_listed_device; mci is indexed by unit. */
int mchap_trace_incongruence_listed_device(int n_list, const int32_t *unit_list_dev, const mchap_unit *units_dev, int steps, int chains,
                                           int burn, const uint64_t *trace_words, int cap, int ploidy_max, double threshold, int32_t *mci,
                                           void *stream);

/* The four summaries above for traces of `words_per_haplotype` 64-bit words per haplotype: 1 (the entry points above), or 2 for a
 * batch that ran on the general sampler (mchap_denovo_trace_words_per_haplotype: units of more than 64 bits per haplotype, ploidies 9
 * to MCHAP_MAX_PLOIDY_DENOVO) -- round 5, so that no shape `assemble` samples needs its traces on the host.  A state is then
 * ploidy x 2 words (a haplotype's words adjacent, genotypes sorted by haplotype as the sampler writes them) and the rows of
 * post_words / mode_words hold ploidy_max x 2 words; a batch launch keeps min(512, mchap_trace_posterior_max_states_wph) states. */
int mchap_trace_posterior_batch_wph_device(int n_units, const mchap_unit *units_dev, int steps, int chains, int burn,
                                           const uint64_t *trace_words, int max_states, int ploidy_max, int words_per_haplotype,
                                           uint64_t *post_words, int32_t *post_counts, int32_t *post_n,
                                           double *mode_stats, int32_t *mode_index, uint64_t *mode_words,
                                           int32_t *mode_count, void *stream);
int mchap_trace_posterior_max_states_wph(int ploidy_max, int words_per_haplotype);
int mchap_trace_posterior_listed_wph_device(int n_list, const int32_t *unit_list_dev, const mchap_unit *units_dev, int steps, int chains,
                                            int burn, const uint64_t *trace_words, int cap, int ploidy_max, int words_per_haplotype,
                                            uint64_t *post_words, int32_t *post_counts, int32_t *post_n, double *mode_stats,
                                            int32_t *mode_index, uint64_t *mode_words, int32_t *mode_count, void *stream);
int mchap_trace_incongruence_batch_wph_device(int n_units, const mchap_unit *units_dev, int steps, int chains, int burn,
                                              const uint64_t *trace_words, int ploidy_max, int words_per_haplotype, double threshold,
                                              int32_t *mci, void *stream);
int mchap_trace_incongruence_listed_wph_device(int n_list, const int32_t *unit_list_dev, const mchap_unit *units_dev, int steps, int chains,
                                               int burn, const uint64_t *trace_words, int cap, int ploidy_max, int words_per_haplotype,
                                               double threshold, int32_t *mci, void *stream);


/* Exact caller for a batch of units that share (n_reads, n_pos, max_allele, n_haps, ploidy), everything resident on the
 * device: what application/call_exact.py:126-179 asks of calling/exact.py per sample, for all samples of a chunk of
 * loci at once.  Every output is optional (NULL = not wanted):
 *   streaming form, float64, no per-genotype array is formed (calling.exact.posterior_mode, exact.py:156-249: a pass
 *   over the genotypes for the normaliser and the mode, a second one for the frequencies):
 *     mode_alleles int64 [U][K], mode_llk / mode_prob / support_prob float64 [U], freqs / occur float64 [U][H]
 *   array form (genotype_likelihoods 266-292: float32 store; genotype_posteriors 295-329 on that float32 array, hence
 *   float32 arithmetic, float64 result) with G = C(H + K - 1, K) genotypes in VCF order:
 *     llks float32 [U][G], llks64 float64 [U][G] (the unrounded values), posteriors float64 [U][G]
 *   and what the caller derives from the posterior array (call_exact.py:142-159; exact.py:332-407): its first maximum
 *   (arr_mode_alleles / arr_mode_prob), the summed probability of the genotypes over the mode's alleles
 *   (arr_support_prob = alternate_dosage_posteriors(...).sum()), posterior_allele_frequencies (arr_freqs, arr_counts,
 *   arr_occur [U][H]).
 * prior: has_prior == 0 -> None; else inbreeding [U] and frequencies [U][H] or NULL.  `workspace` (device,
 * mchap_exact_workspace_bytes) is only needed for the streaming outputs.  Enqueues on `stream`, no synchronisation. */
typedef struct mchap_exact_out {
  int64_t *mode_alleles;
  double *mode_llk, *mode_prob, *support_prob, *freqs, *occur;
  float *llks;
  double *llks64;
  double *posteriors;
  int64_t *arr_mode_alleles;
  double *arr_mode_prob, *arr_support_prob, *arr_freqs, *arr_counts, *arr_occur;
} mchap_exact_out;
int64_t mchap_exact_workspace_bytes(int n_units, int n_haps, int ploidy);
/* ... plus room for llk + log prior of every genotype (U * G float64): the second pass of the streaming form (support,
 * frequencies, occurrence) then reads them back instead of forming every likelihood again (calling/exact.py:156-249 makes
 * both passes from scratch to keep its memory flat; on the device 54 264 genotypes are 434 KB per unit).  Same results;
 * a workspace of either size is accepted. */
int64_t mchap_exact_workspace_bytes_cached(int n_units, int n_haps, int ploidy);
int mchap_exact_call_batch_device(int n_units, const double *reads, int n_reads, int n_pos, int max_allele,
                                  const int64_t *read_counts, const int8_t *haplotypes, int n_haps, int ploidy, int has_prior,
                                  const double *inbreeding, const double *frequencies, const mchap_exact_out *out,
                                  void *workspace, int64_t workspace_bytes, void *stream);

/* Summaries of given posterior arrays [U][G] (calling.exact.posterior_allele_frequencies 332-369, the sum of
 * alternate_dosage_posteriors 372-407 for the array's mode): device pointers, any output may be NULL. */
int mchap_exact_posterior_summaries_batch_device(int n_units, const double *posteriors, int64_t n_genotypes, int ploidy,
                                                 int n_alleles, int64_t *mode_alleles, double *mode_prob, double *support_prob,
                                                 double *freqs, double *counts, double *occur, void *stream);
/* ... and for one array in host memory */
int mchap_exact_posterior_summaries(const double *posteriors, int64_t n_genotypes, int ploidy, int n_alleles,
                                    int64_t *mode_alleles, double *mode_prob, double *support_prob, double *freqs,
                                    double *counts, double *occur);

/* Exact caller, streaming form: replaces calling.exact.posterior_mode (calling/exact.py:156-249) for a batch
 * of units that share (n_reads, n_pos, max_allele, n_haps, ploidy).  Host pointers (one device allocation, copies,
 * mchap_exact_call_batch_device, copies back). */
int mchap_exact_posterior_mode_batch(int n_units, const double *reads, int n_reads, int n_pos, int max_allele,
                                     const int64_t *read_counts, const int8_t *haplotypes, int n_haps, int ploidy,
                                     int has_prior, const double *inbreeding, const double *frequencies,
                                     int64_t *mode_alleles, double *mode_llk, double *mode_prob, double *support_prob,
                                     double *freqs, double *occur);

/* `mchap call`: Gibbs / Metropolis-Hastings sampler over the genotypes of known haplotypes for a batch of units that share
 * a shape: replaces CallingMCMC.fit (calling/classes.py:62-124) -> mcmc_sampler / compound_step / gibbs_options /
 * mh_options / greedy_caller (calling/mcmc.py:15-453) with the Gibbs-step priors of calling/prior.py:30-113.
 *   step_type 0 = "Gibbs", 1 = "Metropolis-Hastings"; initial int64 [U][K] or NULL (greedy_caller);
 *   stream_ids [U]: the unit's Philox stream (results do not depend on batch composition);
 *   genotypes int64 [U][chains][steps][K] (sorted alleles per step), llks float64 [U][chains][steps];
 *   status int32 [U]: 0, or MCHAP_ERR_LIMIT if a chain's table of remembered likelihoods filled up (cannot happen with a
 *   workspace of mchap_call_mcmc_workspace_bytes).
 * The workspace holds, per chain, the counterpart of the reference's llk dict (calling/likelihood.py:36-78), which never
 * forgets an entry.  Device pointers; enqueues on `stream`. */
int64_t mchap_call_mcmc_workspace_bytes(int n_units, int n_haps, int ploidy, int steps, int chains);
/* The same plus the chains' hand-over records (round 5: Gibbs steps at ploidies up to 8 -- a chain that has settled runs the
 * rest of its steps one lane of call_coast_kernel, csrc/call_mcmc_kernel.hpp; 2.2 KB a chain at 16 known haplotypes) and, when
 * n_reads x n_haps x 8 B exceeds the LDS, room for every chain's product table P[r][h] (the sampler then reads it from the
 * workspace: slower, but no limit on the read depth).  mchap_call_mcmc_batch_device needs this many bytes.  Environment
 * MCHAP_HIP_CALL_LANES (measurement): 0 = every chain on its own wavefront to the end, as before round 5; n = chains per
 * wavefront of call_coast_kernel.  The traces are the same either way. */
int64_t mchap_call_mcmc_workspace_bytes_for(int n_units, int n_reads, int n_haps, int ploidy, int steps, int chains);
int mchap_call_mcmc_batch_device(int n_units, const double *reads, int n_reads, int n_pos, int max_allele,
                                 const int64_t *read_counts, const int8_t *haplotypes, int n_haps, int ploidy, int has_prior,
                                 const double *inbreeding, const double *frequencies, const int64_t *initial,
                                 const uint64_t *stream_ids, int steps, int chains, int step_type, uint64_t seed,
                                 int64_t *genotypes, double *llks, int32_t *status, void *workspace, int64_t workspace_bytes,
                                 void *stream);
/* the same with host pointers (one device allocation, copies, the device entry point, copies back) */
int mchap_call_mcmc_batch(int n_units, const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                          const int8_t *haplotypes, int n_haps, int ploidy, int has_prior, const double *inbreeding,
                          const double *frequencies, const int64_t *initial, const uint64_t *stream_ids, int steps, int chains,
                          int step_type, uint64_t seed, int64_t *genotypes, double *llks, int32_t *status);

/* Measurement (bench.py): a timer is a pair of HIP events owned by the caller.  A fit whose cfg.timer is set records them on
 * its own stream right around its sampler launches; mchap_timer_ms waits for the second event and returns the span in
 * milliseconds (< 0: nothing recorded).  No process-wide state: fits on different threads / streams use different timers. */
int mchap_timer_create(void **timer);
double mchap_timer_ms(void *timer);
int mchap_timer_destroy(void *timer);
/* uint64 words per haplotype of this batch's trace_words: 1, or 2 when the batch holds a unit of more than 62 SNVs or more than
 * 64 bits of alleles per haplotype (a pure function of cfg and the units' shapes; < 0: the shapes are beyond the limits) */
int mchap_denovo_trace_words_per_haplotype(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host);
/* name of the sampler kernel(s) a fit of this batch dispatches to (a pure function of cfg and the units' shapes) */
int mchap_denovo_sampler_name(const mchap_denovo_cfg *cfg, int n_units, const mchap_unit *units_host, char *out, int out_len);

/* Host-side helpers of the programs' alignment reader (no GPU involved; the counterpart of the htslib records the reference's
 * io/bam.py:54-229 walks): the records of an INFLATED BAM payload `buf[0 .. n)` from byte `start` (just past the header, or a
 * region's first record) as columns.  mchap_bam_count: the complete records from `start` and the sum of their CIGAR operations.
 * mchap_bam_columns fills, per record: its byte offset, ref_id, pos, end (pos + reference bases its CIGAR consumes), mapq, flag,
 * byte offsets of its packed sequence and of its base qualities, the index of its RG:Z value among the `n_rg` NUL-separated
 * read-group ids `rg_ids` (-1: no such field / unknown id; the fields are walked in order as pysam's get_tag does), qname_id
 * = the index of the first record with the same query name; seg_first [n_records + 1] = first CIGAR operation of each
 * record among the flattened c_* arrays: record, operation code, length, reference / read coordinate at which it starts.
 * Returns MCHAP_OK, or MCHAP_ERR_BAD_ARG for a truncated / malformed record. */
int64_t mchap_bam_count(const uint8_t *buf, int64_t n, int64_t start, int64_t *n_cigar_ops);
int mchap_bam_columns(const uint8_t *buf, int64_t n, int64_t start, int64_t n_records, const char *rg_ids, int n_rg,
                      int64_t *offset, int32_t *ref_id, int32_t *pos, int64_t *end, int32_t *mapq, int32_t *flag,
                      int64_t *seq_off, int64_t *qual_off, int64_t *rg, int64_t *qname_id, int64_t *seg_first,
                      int64_t *c_rec, int64_t *c_op, int64_t *c_len, int64_t *c_ref0, int64_t *c_read0);

/* Introspection */
const char *mchap_version(void);
const char *mchap_last_error(void);
int mchap_device_count(void);
/* bytes of LDS one block of the sampler needs for this shape, or <0 if unsupported */
int64_t mchap_denovo_lds_bytes(int n_reads, int n_pos, int max_allele, int ploidy, int chains, int n_temps);

#ifdef __cplusplus
}
#endif
#endif
