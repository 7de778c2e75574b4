/*
 * mchap_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference algorithm for the north-star hot path
 * (PlantandFoodResearch/MCHap v0.11.1: mchap/assemble/ and mchap/calling/exact.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker / reported CPU baseline -- never as the thing
 * shipped.  Every function cites the reference file:line it follows (paths relative to
 * /root/reference/mchap/).
 *
 * Pinning: the restatement is pinned against vectors captured from the reference
 * itself (imported in the build container with an identity-njit shim) -- see
 * tests/golden/make_golden.py and tests/test_oracle_golden.py -- including whole
 * seeded MCMC traces reproduced step-for-step in ORC_RNG_NUMPY_MT19937 mode.
 */
#ifndef MCHAP_ORACLE_H
#define MCHAP_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_TEMPS 16
#define ORC_MAX_PLOIDY 16
#define ORC_MAX_POS 128  /* (64 until round 4: the wide sampler is checked at 80 / 120 SNVs) */
#define ORC_MAX_ALLELE 8

enum { ORC_RNG_PHILOX = 0, ORC_RNG_NUMPY_MT19937 = 1 };

/* error codes (negative) */
enum {
  ORC_OK = 0,
  ORC_ERR_NAN_LLK = -1,     /* assemble/mcmc.py:330-331 ValueError */
  ORC_ERR_BAD_ARG = -2,     /* AssertionError family */
  ORC_ERR_BREAKS = -3,      /* structural.py:49-50 ValueError */
  ORC_ERR_LIMIT = -4        /* exceeds oracle compile-time limits */
};

/* Sampler configuration: the fields of DenovoMCMC (assemble/mcmc.py:24-40). */
typedef struct {
  int32_t ploidy;
  int32_t steps;
  int32_t chains;
  int32_t n_temps;
  double temperatures[ORC_MAX_TEMPS]; /* ascending, last must be 1.0 */
  double inbreeding;                  /* NaN == None (flat prior) */
  double fix_homozygous;
  double p_recomb;
  double p_partial_dosage;
  double p_dosage;
  int32_t n_intervals;                /* 0 == None: use break_table */
  int32_t llk_cache_threshold;        /* -1 disables; default 100 */
  int32_t rng_kind;
  int32_t reserved;
  uint64_t seed;
  uint64_t stream_id;                 /* philox: unit stream (counter word 3) */
  const double *break_table;          /* [(n_pos+1) x n_pos]; row m = Beta CDF increments for m het bases */
} orc_denovo_cfg;

typedef struct {
  int64_t llk_evals;       /* full likelihood evaluations actually computed */
  int64_t llk_cache_hits;
  int64_t mutation_evals;  /* requested by base_step */
  int64_t structural_evals;/* requested by interval_step */
} orc_stats;

/* ---- likelihood (assemble/likelihood.py) ---- */
double orc_log_likelihood(const double *reads, int n_reads, int n_pos, int max_allele,
                          const int8_t *genotype, int ploidy, const int64_t *read_counts);
double orc_log_likelihood_structural_change(const double *reads, int n_reads, int n_pos, int max_allele,
                                            const int8_t *genotype, int ploidy,
                                            const int8_t *haplotype_indices, int ivl_start, int ivl_stop,
                                            const int64_t *read_counts);

/* ---- priors ---- */
double orc_assemble_log_genotype_prior(const int8_t *dosage, int ploidy, double log_unique_haplotypes, double inbreeding);
double orc_calling_log_genotype_prior(const int64_t *genotype, int ploidy, int64_t unique_haplotypes,
                                      double inbreeding, const double *frequencies /*nullable*/);

/* ---- jitutils ---- */
void orc_get_haplotype_dosage(int8_t *dosage, const int8_t *genotype, int ploidy, int n_base);
int orc_count_haplotype_copies(const int8_t *genotype, int ploidy, int n_base, int h);
void orc_structural_change(int8_t *genotype, int ploidy, int n_base, const int8_t *haplotype_indices, int start, int stop);
void orc_increment_genotype(int64_t *genotype, int ploidy);
int64_t orc_comb_with_replacement(int64_t n, int64_t k);
int64_t orc_genotype_alleles_as_index(const int64_t *alleles, int ploidy);
void orc_index_as_genotype_alleles(int64_t index, int ploidy, int64_t *out);
double orc_add_log_prob(double x, double y);

/* ---- structural option tables (assemble/structural.py) ---- */
void orc_haplotype_segment_labels(const int8_t *genotype, int ploidy, int n_base, int start, int stop, int8_t *labels /*[ploidy][2]*/);
int orc_recombination_step_n_options(const int8_t *labels, int ploidy);
int orc_recombination_step_options(const int8_t *labels, int ploidy, int8_t *options /*[max][ploidy][2]*/);
int orc_dosage_step_n_options(const int8_t *labels, int ploidy);
int orc_dosage_step_options(const int8_t *labels, int ploidy, int8_t *options);

/* ---- deterministic transition vectors (state -> probabilities before random_choice) ---- */
/* mutation.base_step (mutation.py:14-161): fills probs[n_alleles], llks[n_alleles] */
int orc_base_step_probabilities(const double *reads, int n_reads, int n_pos, int max_allele,
                                const int8_t *genotype, int ploidy, double llk, int h, int j, int n_alleles_j,
                                double log_unique_haplotypes, double inbreeding, double temp,
                                const int64_t *read_counts, double *probs, double *llks);
/* structural.interval_step (structural.py:433-587): returns n_options; fills probs[n_options+1], llks[n_options+1],
   option_labels[n_options][ploidy][2] */
int orc_interval_step_probabilities(const double *reads, int n_reads, int n_pos, int max_allele,
                                    const int8_t *genotype, int ploidy, double llk, int start, int stop, int step_type,
                                    double log_unique_haplotypes, double inbreeding, double temp,
                                    const int64_t *read_counts, double *probs, double *llks, int8_t *option_labels);

/* ---- pre-sampling (assemble/snpcalling.py, assemble/mcmc.py:455-541) ---- */
int orc_snp_posterior(const double *read_probs /*[R][A]*/, int n_reads, int max_allele, int n_alleles, int ploidy,
                      double inbreeding, const int64_t *read_counts, double *probs /*[C(n+k-1,k)]*/);
void orc_homozygosity_probabilities(const double *reads, int n_reads, int n_pos, int max_allele, const int8_t *n_alleles,
                                    int ploidy, double inbreeding, const int64_t *read_counts, double *out /*[n_pos][A]*/);
void orc_read_mean_dist(const double *reads, int n_reads, int n_pos, int max_allele, double *dist /*[n_pos][A]*/);

/* ---- the operator: DenovoMCMC.fit (assemble/mcmc.py:103-265) ---- */
int orc_denovo_fit(const orc_denovo_cfg *cfg, const double *reads, int n_reads, int n_pos, int max_allele,
                   const int64_t *read_counts /*nullable*/, const int8_t *n_alleles,
                   const int8_t *initial /*nullable: [chains][ploidy][n_het]*/,
                   int8_t *genotypes_out /*[chains][steps][ploidy][n_pos]*/, double *llks_out /*[chains][steps]*/,
                   orc_stats *stats /*nullable*/);

/* batch of independent units, OpenMP over units (CPU baseline). Units share cfg except stream_id = first_stream + u. */
int orc_denovo_fit_batch(const orc_denovo_cfg *cfg, int n_units, int n_threads,
                         const double *reads /*[U][R][M][A]*/, int n_reads, int n_pos, int max_allele,
                         const int64_t *read_counts /*nullable [U][R]*/, const int8_t *n_alleles /*[M] shared*/,
                         int8_t *genotypes_out /*[U][C][S][K][M]*/, double *llks_out /*[U][C][S]*/, orc_stats *stats);

/* ---- exact caller (calling/exact.py) ---- */
int orc_genotype_likelihoods(const double *reads, int n_reads, int n_pos, int max_allele, int ploidy,
                             const int8_t *haplotypes, int n_haps, const int64_t *read_counts,
                             float *out /*[G] float32 as the reference stores*/, double *out64 /*nullable fp64*/);
int orc_genotype_posteriors_f32(const float *llks, int64_t n_genotypes, int ploidy, int n_alleles,
                                int has_prior, double inbreeding, const double *frequencies, double *out);
int orc_genotype_posteriors_f64(const double *llks, int64_t n_genotypes, int ploidy, int n_alleles,
                                int has_prior, double inbreeding, const double *frequencies, double *out);
int orc_posterior_mode(const double *reads, int n_reads, int n_pos, int max_allele, int ploidy,
                       const int8_t *haplotypes, int n_haps, const int64_t *read_counts,
                       int has_prior, double inbreeding, const double *frequencies,
                       int64_t *mode_alleles, double *mode_llk, double *mode_prob, double *support_prob /*nullable*/,
                       double *freqs /*nullable [H]*/, double *occur /*nullable [H]*/);
void orc_posterior_allele_frequencies_f64(const double *posteriors, int64_t n_genotypes, int ploidy, int n_alleles,
                                          double *freqs, double *counts, double *occur);

/* ---- `mchap call`: Gibbs / Metropolis-Hastings sampler over known haplotypes (calling/mcmc.py, calling/classes.py:14-124,
 * calling/prior.py:30-113, calling/likelihood.py:8-78) ---- */
/* calling/prior.py:30-113: Gibbs-step prior of the variable allele (has_prior == 0: the flat prior of 30-52) */
double orc_log_genotype_allele_prior(const int64_t *genotype, int ploidy, int variable_allele, int64_t unique_haplotypes,
                                     int has_prior, double inbreeding, const double *frequencies /*nullable*/);
/* calling/mcmc.py:15-229: transition vectors of one sub-step (step_type 0 Gibbs, 1 Metropolis-Hastings), no cache */
int orc_call_step_options(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                          const int8_t *haplotypes, int n_haps, const int64_t *genotype, int ploidy, int variable_allele,
                          int step_type, int has_prior, double inbreeding, const double *frequencies,
                          double *llks /*[H]*/, double *lpriors /*[H]*/, double *probs /*[H]*/);
/* calling/mcmc.py:393-453 */
int orc_greedy_caller(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                      const int8_t *haplotypes, int n_haps, int ploidy, int has_prior, double inbreeding,
                      const double *frequencies, int64_t *genotype_out);
/* CallingMCMC.fit (calling/classes.py:62-124) with mcmc_sampler (calling/mcmc.py:330-390, cache=True: a likelihood is
 * remembered per SORTED genotype, i.e. a later request for the same alleles in another order gets the first value) */
int orc_call_mcmc(const double *reads, int n_reads, int n_pos, int max_allele, const int64_t *read_counts,
                  const int8_t *haplotypes, int n_haps, int ploidy, int has_prior, double inbreeding,
                  const double *frequencies, int steps, int chains, int step_type, const int64_t *initial /*nullable*/,
                  int rng_kind, uint64_t seed, uint64_t stream_id,
                  int64_t *genotypes_out /*[chains][steps][ploidy]*/, double *llks_out /*[chains][steps]*/);

/* ---- RNG test hooks ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_philox_double(uint64_t seed, uint64_t stream_id, uint32_t substream, uint64_t n);
uint32_t orc_philox_interval(uint64_t seed, uint64_t stream_id, uint32_t substream, uint64_t n, uint32_t max);
void orc_mt_doubles(uint32_t seed, int n, double *out);

const char *orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
