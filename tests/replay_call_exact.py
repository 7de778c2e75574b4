"""Test-side helpers around mchap_amd.application / mchap_amd.io (the replay of the reference's golden VCFs):
the oracle behind the function names of mchap_amd.calling, so that the same plumbing can be driven by the CPU
oracle (tests/test_oracle_golden.py) as well as by the GPU exact caller (tests/test_gpu_call_exact_goldens.py).
Data under tests/golden/reference_data are the reference's own test files."""
from mchap_amd.application import call_exact, call_exact_record, sample_reads as _sample_reads  # noqa: F401
from mchap_amd.io import (DenovoLocus, Locus, extract_read_variants, qual_of_prob, read_bam, read_bed4, read_vcf,  # noqa: F401
                          vcfstr)


class OracleBackend:
    """The CPU oracle behind the function names of mchap_amd.calling (the enumeration helpers are host code there)."""

    def __init__(self):
        from mchap_amd import calling as host
        from oracle import binding as orc

        self.orc = orc
        self.index_as_genotype_alleles = host.index_as_genotype_alleles
        self.alternate_dosage_posteriors = host.alternate_dosage_posteriors

    def genotype_likelihoods(self, reads, ploidy, haplotypes, read_counts=None):
        return self.orc.genotype_likelihoods(reads, ploidy, haplotypes, read_counts)[0]  # the float32 array (exact.py:254)

    def genotype_posteriors(self, llks, ploidy, n_alleles, prior=None):
        return self.orc.genotype_posteriors(llks, ploidy, n_alleles, prior)

    def posterior_allele_frequencies(self, probs, ploidy, n_alleles):
        return self.orc.posterior_allele_frequencies(probs, ploidy, n_alleles)

    def posterior_mode(self, reads, ploidy, haplotypes, read_counts=None, prior=None, **kw):
        return self.orc.posterior_mode(reads, ploidy, haplotypes, read_counts, prior)


def call_record(rec, bams, sample_names, calling=None, **kw):
    """-> {sample: column} (the sample columns of one record)."""
    return call_exact_record(rec, bams, sample_names, calling=calling, **kw)[3]


def sample_reads(locus, bam, sample, error_rate=0.0024):
    """-> (calls, distinct read distributions, counts)"""
    sr = _sample_reads(locus, [(sample, bam)], error_rate)
    return sr["calls"], sr["dists"], sr["counts"]
