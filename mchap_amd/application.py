"""Application layer over the kernels: the per-locus plumbing of `mchap call-exact` and `mchap assemble`
(reference application/baseclass.py:140-302, application/call_exact.py:52-199, application/assemble.py:95-176) from
BAM / VCF / BED files to VCF record lines, without pysam.  The compute goes through `mchap_amd.calling` (exact caller
kernels) and `mchap_amd.DenovoMCMC` (sampler kernels); everything else here is host-side formatting.

Parity: the reference's RNG-free `call-exact` golden VCFs are reproduced line for line (tests/test_gpu_call_exact_goldens.py).
"""
import numpy as np

from contextlib import nullcontext as _nullcontext

from . import encoding
from .io import DenovoLocus, Locus, extract_read_variants, qual_of_prob, read_alignments, read_bam, read_bed4, read_vcf, vcfstr  # noqa: F401

SAMPLE_FIELDS = ("GT", "GQ", "SQ", "DP", "RCOUNT", "RCALLS", "MEC", "MECP", "GPM", "SPM", "MCI")


READ_FILTER = dict(min_quality=20, skip_duplicates=True, skip_qcfail=True, skip_supplementary=True)


class ReadSource:
    """The alignment files of a run, each read once, and the read encoding settings (application/baseclass.py:48-57).
    sample_bams: ordered {sample: path}, or {pool: [(sample, path), ...]} when samples are pooled (--sample-pool: a pool's
    reads are those of its samples, concatenated: baseclass.py:153-187)."""

    def __init__(self, sample_bams, error_rate=0.0024, use_phred=False, read_group_field="SM", mapping_quality=20,
                 skip_duplicates=True, skip_qcfail=True, skip_supplementary=True, workers=1):
        self.pools = {k: ([(k, v)] if isinstance(v, str) else list(v)) for k, v in sample_bams.items()}
        self.samples = list(self.pools)
        self.error_rate, self.use_phred = error_rate, use_phred
        self.filter = dict(min_quality=mapping_quality, skip_duplicates=skip_duplicates, skip_qcfail=skip_qcfail,
                           skip_supplementary=skip_supplementary)
        paths = list(dict.fromkeys(p for pairs in self.pools.values() for _, p in pairs))
        self.workers = int(workers)
        self.id_field = read_group_field

        def load(q):
            # BAM: block-wise reader with columnar records (and region fetches through the .bai index for big files);
            # SAM text: the record reader
            with open(q, "rb") as f:
                magic = f.read(2)
            if magic == b"\x1f\x8b":
                from .io import BamFile

                return BamFile(q, read_group_field, workers=self.workers)
            return read_alignments(q, read_group_field)

        if workers > 1 and len(paths) > 1:
            # --cores: the files are read by a pool of threads (zlib releases the interpreter lock)
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=int(workers)) as ex:
                loaded = list(ex.map(load, paths))
        else:
            loaded = [load(q) for q in paths]
        self.bams = dict(zip(paths, loaded))

    WHOLE_FILE_BYTES = 256 << 20  # smaller files are inflated once; bigger ones are fetched region by region (needs the .bai)

    def reads(self, locus, sample):
        from .io import BamFile, extract_read_variants_columns

        M = len(locus.positions)
        parts = []
        for name, path in self.pools[sample]:
            bam = self.bams[path]
            if isinstance(bam, BamFile):
                region = bam.index is not None and len(bam.data) > self.WHOLE_FILE_BYTES
                cols = bam.columns(locus.contig, locus.start, locus.stop) if region else bam.columns()
                parts.append(extract_read_variants_columns(locus, cols, name, as_codes=True, **self.filter))
            else:
                c, q = extract_read_variants(locus, bam, name, **self.filter)
                parts.append((_codes(c), q))
        if parts:
            chars, quals = np.concatenate([c for c, _ in parts]), np.concatenate([q for _, q in parts])
        else:
            chars, quals = np.empty((0, M), dtype=np.uint8), np.empty((0, M), dtype=np.int16)
        return encode_reads(locus, chars, quals, self.error_rate, self.use_phred)


class MatrixSource:
    """Reads given as what extract_read_variants yields -- {(target name, sample): (chars [n_reads, n_snv] as str or uint8
    ASCII codes, summed base qualities [n_reads, n_snv])} -- instead of alignment files: a caller that has its pileups
    already (and the docs/example fixture of the tests and of bench.py's extra.config1)."""

    def __init__(self, samples, matrices, error_rate=0.0024, use_phred=False):
        self.samples = list(samples)
        self.matrices = matrices
        self.error_rate, self.use_phred = error_rate, use_phred
        self.bams = {}
        self._codes = {}

    def codes(self, name, sample):
        """(characters as uint8 ASCII codes, qualities as int16) of a (target, sample): converted once."""
        key = (name, sample)
        got = self._codes.get(key)
        if got is None:
            chars, quals = self.matrices[key]
            got = self._codes[key] = (_codes(chars), np.asarray(quals, dtype=np.int16))
        return got

    def reads(self, locus, sample):
        chars, quals = self.matrices[(locus.name, sample)]
        return encode_reads(locus, np.asarray(chars), np.asarray(quals, dtype=np.int16), self.error_rate, self.use_phred)


def load_matrices(path):
    """A pileup file written as numpy .npz (tests/golden/make_example_fixture.py documents the layout: samples, contigs,
    targets, per target its SNV positions and alleles, per (target, sample) the character matrix and the qualities) ->
    (samples, targets [(contig, start, stop, name)], variant records, {(target name, sample): (chars, quals)},
    contigs [(name, length)])."""
    z = np.load(path)
    samples = [str(x) for x in z["samples"]]
    targets = [(str(c), int(a), int(b), str(n)) for c, a, b, n in zip(z["target_contig"], z["target_start"], z["target_stop"], z["target_name"])]
    variants, matrices = [], {}
    for li, (contig, start, stop, name) in enumerate(targets):
        for p, al in zip(z["pos_%d" % li], z["alleles_%d" % li]):
            al = str(al)
            variants.append(dict(chrom=contig, pos=int(p) + 1, id=".", ref=al[0], alts=tuple(al[1:]), info={}))
        for si, s_ in enumerate(samples):
            matrices[(name, s_)] = (z["chars_%d_%d" % (li, si)], z["quals_%d_%d" % (li, si)])
    contigs = [(str(c), int(n)) for c, n in zip(z["contigs"], z["contig_lengths"])]
    return samples, targets, variants, matrices, contigs


def sample_reads(locus, pairs, error_rate=0.0024, use_phred=False, read_filter=None):
    """encode_sample_reads (application/baseclass.py:140-210) for one sample (or pool): pairs = [(read-group sample name,
    alignment as read_alignments returns it), ...] -> dict(chars, calls, depth, dists (distinct rows), counts)."""
    M = len(locus.positions)
    parts = [extract_read_variants(locus, bam, name, **(read_filter or READ_FILTER)) for name, bam in pairs]
    if parts:
        chars, quals = np.concatenate([c for c, _ in parts]), np.concatenate([q for _, q in parts])
    else:
        chars, quals = np.empty((0, M), dtype=np.uint8), np.empty((0, M), dtype=np.int16)
    return encode_reads(locus, chars, quals, error_rate, use_phred)


def _codes(chars):
    """A character matrix ('U1' strings, or uint8 ASCII codes already) as uint8 ASCII codes."""
    chars = np.asarray(chars)
    if chars.dtype == np.uint8:
        return chars
    return chars.astype("S1").view(np.uint8).reshape(chars.shape) if chars.size else np.empty(chars.shape, dtype=np.uint8)


def encode_reads(locus, chars, quals, error_rate=0.0024, use_phred=False):
    """Character matrix (strings or ASCII codes) + qualities -> allele calls, probabilistic rows, de-duplicated rows with
    counts, depth (application/baseclass.py:189-207)."""
    M = len(locus.positions)
    chars = _codes(chars)
    calls = np.full(chars.shape, -1, dtype=np.int8)
    if M and chars.shape[0]:
        # allele index of every character: one table look-up per SNV column
        lut = np.full((M, 256), -1, dtype=np.int8)
        for j in range(M):
            for a, c in enumerate(locus.alleles[j]):
                if lut[j, ord(c)] < 0:
                    lut[j, ord(c)] = a
        calls = lut[np.arange(M)[None, :], chars]
    if use_phred or M == 0 or calls.shape[0] == 0:
        dists = encoding.encode_read_distributions(locus.n_alleles, calls, quals if use_phred else None, error_rate=error_rate)
        uniq, counts = encoding.unique_counts(dists)
    else:
        # qualities ignored: a row of distributions is a function of its row of calls, so the distinct rows (in order of first
        # appearance, with their counts: mset.unique_counts at application/baseclass.py:207) are found on the int8 calls and
        # only those are turned into distributions
        ucalls, counts = encoding.unique_counts(np.ascontiguousarray(calls))
        uniq = encoding.encode_read_distributions(locus.n_alleles, ucalls, None, error_rate=error_rate)
    depth = (chars != ord("-")).sum(axis=0) if M else np.array([])
    return dict(chars=chars, calls=calls, depth=depth, dists=uniq, counts=counts)


def resolve_seed(seed):
    """The run's random seed: the reference's programs default to 42 (application/arguments.py:538-546) and give every
    unit the same value (baseclass.py:360-388, assemble/mcmc.py:140-142).  None (library use) draws one value, once."""
    if seed is None:
        return int(np.random.randint(0, 2 ** 31 - 1))
    return int(seed)


def device_unit_budget(bytes_per_unit, fraction=0.5, least=1, most=1 << 20, reserve=0):
    """How many units of `bytes_per_unit` device bytes a launch may hold: a share of the free HBM (the reference streams
    record by record; here a file is processed in blocks of records, each block one launch per shape).  `reserve`: bytes
    the launches take whatever the number of units (at most three quarters of the free memory are set aside for them)."""
    try:
        import torch

        free, _ = torch.cuda.mem_get_info()
    except Exception:
        free = 8 << 30
    free = max(free - int(reserve), free // 4)
    return int(max(least, min(most, (free * fraction) // max(1, int(bytes_per_unit)))))


def _mec(calls, genotype):
    """minimum_error_correction summed over reads (encoding/integer/stats.py:18-39)."""
    if len(calls) == 0:
        return 0
    diff = (calls[:, None, :] != genotype[None]) & (calls[:, None, :] >= 0)
    return int(diff.sum(axis=-1).min(axis=-1).sum())


def _exact_units(records, source, samples, prior_tag, allele_filter=None):
    """Everything `call-exact` needs from the input side, record by record: the locus, per sample the encoded reads,
    and which (record, sample) units need the kernel (a record with a single haplotype, or no variable position, has
    one genotype with probability 1; an invalid record is not called at all)."""
    out = []
    for rec in records:
        locus = Locus(rec, prior_tag, allele_filter)
        H, M = locus.haplotypes.shape
        invalid = None
        if locus.mask_reference_allele and H == 1:
            invalid = "NOA"
        elif np.any(np.isnan(locus.frequencies)):
            invalid = "AF0"
        per = {s: source.reads(locus, s) for s in samples}
        out.append(dict(rec=rec, locus=locus, invalid=invalid, reads=per, needs_kernel=(invalid is None and M > 0 and H > 1)))
    return out


def _run_exact_groups(units, ploidy_of, inbreeding_of, full, backend=None):
    """The exact caller for all units that need it, grouped by shape (positions, alleles, haplotypes, ploidy): one device
    call per group, reads padded to the group's deepest sample with weight-0 gap rows.  `backend`: an object with the
    reference's module-level functions (the oracle replay of the CPU tests); None = the device batch.
    Returns {(record index, sample): dict(alleles, gprob, sprob, freqs, occur[, GL, GP])}."""
    from .device import ExactDeviceBatch

    groups = {}
    for ri, unit in enumerate(units):
        if not unit["needs_kernel"]:
            continue
        locus = unit["locus"]
        H, M = locus.haplotypes.shape
        for s, sr in unit["reads"].items():
            A = sr["dists"].shape[2] if sr["dists"].ndim == 3 and sr["dists"].shape[0] else int(max(locus.n_alleles))
            groups.setdefault((M, A, H, int(ploidy_of(s))), []).append((ri, s))
    results = {}
    for (M, A, H, K), members in groups.items():
        if backend is not None:
            for ri, s in members:
                unit = units[ri]
                sr, locus = unit["reads"][s], unit["locus"]
                F = inbreeding_of(s)
                prior = None if F is None else (F, locus.frequencies)
                results[(ri, s)] = _exact_one(backend, sr, locus.haplotypes, K, prior, full)
            continue
        Rmax = max(max(len(units[ri]["reads"][s]["dists"]), 1) for ri, s in members)
        from math import comb

        G = comb(H + K - 1, K)
        # device bytes per unit: reads + haplotypes + (array form) the float32 / float64 genotype arrays, or (streaming
        # form) the joint values kept between the passes -- the group is cut into chunks that fit the free HBM
        per_unit = Rmax * M * A * 8 + Rmax * 8 + H * M + (G * 20 if full else G * 8) + 4096
        step = device_unit_budget(per_unit)
        if len(members) > step:
            for c0 in range(0, len(members), step):
                sub = {}
                part = members[c0:c0 + step]
                keep = sorted({ri for ri, _ in part})
                remap = {ri: i for i, ri in enumerate(keep)}
                sub_units = [dict(units[ri], reads={s: units[ri]["reads"][s] for r2, s in part if r2 == ri}) for ri in keep]
                sub = _run_exact_groups(sub_units, ploidy_of, inbreeding_of, full, backend)
                for (i, s), v in sub.items():
                    results[(keep[i], s)] = v
            continue
        U = len(members)
        reads = np.full((U, Rmax, M, A), np.nan)
        counts = np.zeros((U, Rmax), dtype=np.int64)
        haps = np.zeros((U, H, M), dtype=np.int8)
        Fs = np.zeros(U)
        frs = np.zeros((U, H))
        has_prior = inbreeding_of(members[0][1]) is not None
        for i, (ri, s) in enumerate(members):
            sr, locus = units[ri]["reads"][s], units[ri]["locus"]
            n = len(sr["dists"])
            if n:
                reads[i, :n] = sr["dists"]
                counts[i, :n] = sr["counts"]
            haps[i] = locus.haplotypes
            if has_prior:
                Fs[i] = inbreeding_of(s)
                frs[i] = locus.frequencies
        try:
            batch = ExactDeviceBatch(reads, K, haps, counts, (Fs, frs) if has_prior else None)
            batch.run(streaming=not full, arrays=full)
        except NotImplementedError as e:
            # a shape the library does not take (more than 2^62 genotypes, ploidy above 15, tables beyond the LDS): the records
            # of this group are written with null genotypes and FILTER=LIMIT, the file goes on (round 5: used to raise mid-file)
            _limit_units(units, members, "call-exact", "%d haplotypes x ploidy %d: %s" % (H, K, e))
            continue
        if full:
            arr = batch.array_results(True)
            for i, key in enumerate(members):
                results[key] = dict(alleles=arr["alleles"][i], gprob=arr["prob"][i], sprob=arr["support_prob"][i], freqs=arr["freqs"][i],
                                    occur=arr["occur"][i], GL=arr["llks"][i].astype(np.float64) / np.log(10), GP=arr["posteriors"][i])
        else:
            al, _, gp, sp, fq, oc = batch.mode_results()
            for i, key in enumerate(members):
                results[key] = dict(alleles=al[i], gprob=gp[i], sprob=sp[i], freqs=fq[i], occur=oc[i])
    return results


def _limit_units(units, members, program, reason):
    """Marks the records of `members` [(record index, sample)] as beyond a limit of the library: FILTER=LIMIT, null genotypes."""
    import sys

    for ri in sorted({ri for ri, _ in members}):
        if units[ri].get("invalid") is None:
            units[ri]["invalid"] = "LIMIT"
            rec = units[ri]["rec"]
            sys.stderr.write("mchap_amd %s: record %s:%s not called (written with FILTER=LIMIT): %s\n" % (
                program, rec.get("chrom", "?"), rec.get("pos", "?"), reason))


def _exact_one(calling, sr, haps, ploidy, prior, full):
    """One unit through the reference's own sequence of calls (application/call_exact.py:126-179) on `calling`."""
    H = len(haps)
    if full:
        llks = calling.genotype_likelihoods(sr["dists"], ploidy, haps, read_counts=sr["counts"])
        probs = calling.genotype_posteriors(llks, ploidy, H, prior=prior)
        idx = int(np.argmax(probs))
        alleles = calling.index_as_genotype_alleles(idx, ploidy)
        sprob = calling.alternate_dosage_posteriors(alleles, probs)[1].sum()
        freqs, _, occur = calling.posterior_allele_frequencies(probs, ploidy, H)
        return dict(alleles=alleles, gprob=probs[idx], sprob=sprob, freqs=freqs, occur=occur,
                    GL=llks.astype(np.float64) / np.log(10), GP=probs)
    alleles, _, gprob, sprob, freqs, occur = calling.posterior_mode(
        sr["dists"], ploidy, haps, read_counts=sr["counts"], prior=prior, return_support_prob=True,
        return_posterior_frequencies=True, return_posterior_occurrence=True)
    return dict(alleles=alleles, gprob=gprob, sprob=sprob, freqs=freqs, occur=occur)


def _per_sample(value, samples):
    """A scalar, or a {sample: value} mapping (the reference's per-sample ploidy / inbreeding files), as a function."""
    if isinstance(value, dict):
        return lambda s: value[s]
    return lambda s: value


def _format_exact_record(unit, samples, results, ri, ploidy_of, report, prior_tag):
    """(FILTER, INFO, FORMAT, {sample: column}) of one record from the units' results (application/baseclass.py:220-302,
    call_exact.py:84-199)."""
    from .vcfheader import report_fields

    info_opt, fmt_opt = report_fields(report)
    rec, locus, invalid = unit["rec"], unit["locus"], unit["invalid"]
    haps = locus.haplotypes
    H, M = haps.shape
    cols, gts = {}, {}
    acp_sum = np.zeros(H)
    aop_not = np.ones(H)
    aop_sum = np.zeros(H)
    snvdp_sum = np.zeros(M)
    dps, rcounts = [], []
    nan_arrays = False
    ploidy_total = 0
    for sample in samples:
        ploidy = int(ploidy_of(sample))
        ploidy_total += ploidy
        sr = unit["reads"][sample]
        calls, depth = sr["calls"], sr["depth"]
        rcount = len(calls)
        dp = np.round(np.mean(depth)) if len(depth) else np.nan
        rcalls = int((calls >= 0).sum())
        dps.append(dp)
        rcounts.append(rcount)
        if M:
            snvdp_sum += np.round(depth)
        if invalid:
            gts[sample] = np.full(ploidy, -1, int)
            fields = ["/".join(["."] * ploidy), ".", ".", vcfstr(float(dp)), str(rcount), str(rcalls), ".", ".", ".", ".", "."]
            fields += ["."] * len(fmt_opt)
            cols[sample] = ":".join(fields)
            nan_arrays = True
            continue
        if M == 0 or H == 1:
            # a single haplotype: one genotype with probability 1 (the reference's arithmetic gives exactly that)
            res = dict(alleles=np.zeros(ploidy, int), gprob=1.0, sprob=1.0, freqs=np.ones(1), occur=np.ones(1), GL=np.zeros(1), GP=np.ones(1))
        else:
            res = results[(ri, sample)]
        alleles = np.asarray(res["alleles"])
        gprob, sprob = res["gprob"], res["sprob"]
        gts[sample] = alleles
        freqs, occur = np.asarray(res["freqs"], float), np.asarray(res["occur"], float)
        acp_sum += freqs * ploidy
        aop_sum += occur
        aop_not *= 1 - occur
        mec = _mec(calls, haps[alleles])
        denom = int((calls >= 0).sum())
        mecp = mec / denom if denom > 0 else np.nan
        fields = ["/".join(str(a) for a in alleles), vcfstr(qual_of_prob(gprob)), vcfstr(qual_of_prob(sprob)), vcfstr(float(dp)),
                  str(rcount), str(rcalls), str(mec), vcfstr(float(mecp)), vcfstr(float(gprob)), vcfstr(float(sprob)), "."]
        for tag in fmt_opt:
            if tag == "AFP":
                fields.append(vcfstr(freqs))
            elif tag == "ACP":
                fields.append(vcfstr(freqs * ploidy))
            elif tag == "AOP":
                fields.append(vcfstr(occur))
            elif tag == "SNVDP":
                fields.append(vcfstr(np.round(depth).astype(float)) if len(depth) else ".")
            elif tag in ("GL", "GP"):
                fields.append(vcfstr(np.asarray(res[tag], float)))
        cols[sample] = ":".join(fields)
    # ---- record level ----
    counts = np.zeros(H, int)
    for g in gts.values():
        for a in g:
            if a >= 0:
                counts[a] += 1
    info = [("AN", int(counts.sum())), ("UAN", int((counts > 0).sum())), ("AC", counts[1:])]
    if locus.mask_reference_allele:
        info.append(("REFMASKED", True))
    info += [("NS", sum(int(np.any(g >= 0)) for g in gts.values())), ("MCI", 0),
             ("DP", float(np.nansum(dps)) if M else np.nan), ("RCOUNT", int(np.nansum(rcounts))), ("END", locus.stop),
             ("NVAR", M), ("SNVPOS", np.array(locus.positions, int) - locus.start + 1)]
    null_r = np.full(H, np.nan)
    for tag in info_opt:
        if tag == "AFPRIOR":
            info.append(("AFPRIOR", locus.frequencies))
        elif tag == "ACP":
            info.append(("ACP", null_r if nan_arrays else acp_sum))
        elif tag == "AFP":
            info.append(("AFP", null_r if nan_arrays else acp_sum / ploidy_total))
        elif tag == "AOP":
            info.append(("AOP", null_r if nan_arrays else 1 - aop_not))
        elif tag == "AOPSUM":
            info.append(("AOPSUM", null_r if nan_arrays else aop_sum))
        elif tag == "SNVDP":
            info.append(("SNVDP", snvdp_sum))
    parts = []
    for k, v in info:
        if isinstance(v, bool):
            if v:
                parts.append(k)
        else:
            parts.append("%s=%s" % (k, vcfstr(np.asarray(v) if isinstance(v, np.ndarray) else v)))
    fmt = ":".join(SAMPLE_FIELDS + tuple(fmt_opt))
    return invalid or "PASS", ";".join(parts), fmt, cols


def _source(sample_bams, base_error_rate, use_base_phred_scores, read_kw):
    if hasattr(sample_bams, "reads") and hasattr(sample_bams, "samples"):  # a ReadSource / MatrixSource
        return sample_bams
    return ReadSource(sample_bams, error_rate=base_error_rate, use_phred=use_base_phred_scores, **(read_kw or {}))


def call_exact_record(rec, bams, samples, ploidy=4, report=(), error_rate=0.0024, use_phred=False, prior_tag=None,
                      inbreeding=None, calling=None):
    """One record of the input VCF -> (FILTER, INFO string, FORMAT string, {sample: column}) as `mchap call-exact` writes
    them.  bams: {sample: alignment as read_alignments returns it}.  `calling`: None = the device batch; or an object with
    the reference's functions (test replays)."""
    source = ReadSource({}, error_rate=error_rate, use_phred=use_phred)
    source.pools = {s: [(s, s)] for s in samples}
    source.bams = {s: bams[s] for s in samples}
    units = _exact_units([rec], source, samples, prior_tag)
    _, fmt_opt = __import__("mchap_amd.vcfheader", fromlist=["report_fields"]).report_fields(report)
    full = ("GL" in fmt_opt) or ("GP" in fmt_opt)
    ploidy_of, inbreeding_of = _per_sample(ploidy, samples), _per_sample(inbreeding, samples)
    results = _run_exact_groups(units, ploidy_of, inbreeding_of, full, calling)
    return _format_exact_record(units[0], samples, results, 0, ploidy_of, report, prior_tag)


def _allele_filter(vcf_path, text):
    """--filter-input-haplotypes text -> what io.Locus takes (None: no filter)."""
    if text is None:
        return None
    from .io import parse_allele_filter, vcf_info_numbers

    field, compare, number = parse_allele_filter(text)
    numbers = vcf_info_numbers(vcf_path)
    if field not in numbers:
        raise ValueError("Allele filter field not found in header '%s'" % field)
    return field, compare, number, numbers[field]


def _blocks(items, n):
    for i in range(0, len(items), max(1, n)):
        yield items[i:i + max(1, n)]


def call_exact(vcf_path, sample_bams, ploidy=4, report=(), base_error_rate=0.0024, use_base_phred_scores=False,
               prior_frequencies_tag=None, inbreeding=None, calling=None, filter_input_haplotypes=None, read_kw=None,
               records_per_block=4096, shard=None):
    """`mchap call-exact` over a VCF of known haplotypes: yields one VCF record line per input record (no header).
    sample_bams: ordered mapping sample name -> BAM path (or pool -> [(sample, path)], or a ReadSource); ploidy /
    inbreeding: a value or {sample: value}.  The records are processed in blocks: the (record x sample) units of a block
    are encoded, grouped by shape and evaluated in one device call per shape (cut further when a group would not fit the
    free HBM), and the block's lines are yielded before the next block is read: host and device memory stay bounded by
    the block, as with the reference's record-by-record stream.  shard = (rank, world): this process's contiguous share
    of the records (mchap_amd.shard.shard_range)."""
    from .vcfheader import report_fields

    source = _source(sample_bams, base_error_rate, use_base_phred_scores, read_kw)
    samples = source.samples
    _, records = read_vcf(vcf_path)
    records = _shard(records, shard)
    report = tuple(report)
    full = bool({"GL", "GP"} & set(report_fields(report)[1]))
    ploidy_of, inbreeding_of = _per_sample(ploidy, samples), _per_sample(inbreeding, samples)
    allele_filter = _allele_filter(vcf_path, filter_input_haplotypes)
    for block in _blocks(records, records_per_block):
        units = _exact_units(block, source, samples, prior_frequencies_tag, allele_filter)
        results = _run_exact_groups(units, ploidy_of, inbreeding_of, full, calling)
        for ri, unit in enumerate(units):
            rec = unit["rec"]
            flt, info, fmt, cols = _format_exact_record(unit, samples, results, ri, ploidy_of, report, prior_frequencies_tag)
            alts = unit["locus"].sequences[1:]  # (the input's, minus the alleles --filter-input-haplotypes removed)
            alt = ",".join(alts) if alts else "."
            yield "\t".join([rec["chrom"], str(rec["pos"]), rec["id"], rec["ref"], alt, ".", flt, info, fmt] + [cols[s] for s in samples])


def _shard(items, shard):
    """This rank's contiguous share of the items (targets or records): loci are independent, so the programs shard the
    work list before anything touches the GPU (application/baseclass.py:360-388 deals blocks of loci to its workers)."""
    if shard is None:
        return items
    from .shard import shard_range

    rank, world = shard
    lo, hi = shard_range(len(items), rank, world)
    return items[lo:hi]


# ---------------------------------------------------------------------------------------------------------
# mchap assemble (application/assemble.py:95-252, assemble/haplotype_calling.py:4-64)
# ---------------------------------------------------------------------------------------------------------
def call_posterior_haplotypes(posteriors, threshold=0.01):
    """The haplotype alleles of a VCF record from the samples' posteriors (same name and result as the reference's
    assemble/haplotype_calling.py): a haplotype is listed when its posterior probability of occurrence reaches
    `threshold` in at least one sample; listed haplotypes are ranked by their expected dosage summed over the samples
    that list them, the all-reference haplotype always first.  Returns (haplotypes int8 [n, n_base], ref_observed)."""
    from .classes import _first_occurrence_unique

    n_base = posteriors[0].genotypes.shape[-1]
    rows, weights = [], []
    for post in posteriors:
        haps, dosage, occurrence = post.allele_frequencies(dosage=True)
        listed = occurrence >= threshold
        rows.append(np.asarray(haps[listed], dtype=np.int8).reshape(int(np.sum(listed)), n_base))
        weights.append(dosage[listed])
    rows = np.concatenate(rows) if rows else np.zeros((0, n_base), np.int8)
    weights = np.concatenate(weights) if weights else np.zeros(0)
    if len(rows):
        first, ids = _first_occurrence_unique(rows)      # distinct haplotypes in order of first listing
        distinct = rows[first]
        score = np.zeros(len(distinct))
        np.add.at(score, ids, weights)                   # summed sample after sample
    else:
        distinct, score = rows, np.zeros(0)
    is_ref = (distinct == 0).all(axis=1) if len(distinct) else np.zeros(0, bool)
    ref_observed = bool(is_ref.any())
    distinct, score = distinct[~is_ref], score[~is_ref]
    haplotypes = np.concatenate([distinct, np.zeros((1, n_base), np.int8)]).astype(np.int8)
    value = np.append(score, (score.max() if len(score) else -1.0) + 1.0)  # the reference allele outranks everything
    return haplotypes[np.flip(np.argsort(value))], ref_observed


def _genotype_posterior_array(post, labels, ploidy):
    """Posterior probabilities of all genotypes over the record's labelled alleles, in VCF order (application/assemble.py:
    276-305): a posterior genotype that holds an allele the record does not list cannot be encoded and is left out."""
    from math import comb

    from .calling_mcmc import _vcf_index

    # (every allele of the record counts, a masked reference allele included: Number=G.  The reference sizes the array by
    # len(labels), one short when the reference allele is masked, and then fails with an IndexError on the last allele.)
    n_alleles = max(labels.values()) + 1 if labels else 1
    out = np.zeros(comb(n_alleles + ploidy - 1, ploidy), float)
    for haps, prob in zip(post.genotypes, post.probabilities):
        alleles = np.sort([labels.get(np.asarray(h, dtype=np.int8).tobytes(), -1) for h in haps])
        if alleles[0] >= 0:
            out[int(_vcf_index(alleles[None])[0])] = prob
    return out


def _limit_record_line(locus, samples, report=(), ploidy_of=None):
    """The record of a target the library could not sample (beyond its shape limits: the reference has none, so this is a gap
    of this build, made visible): REF, no ALT, FILTER = LIMIT, the INFO fields that do not depend on the sampler, and null
    genotypes of each sample's ploidy -- never an omitted record (an incomplete file is worse than a flagged one)."""
    from .vcfheader import report_fields

    _, fmt_opt = report_fields(report)
    M = len(locus.positions)
    info = "AN=0;UAN=0;AC=.;NS=0;MCI=0;DP=.;RCOUNT=.;END=%d;NVAR=%d;SNVPOS=%s" % (
        locus.stop, M, vcfstr(np.array(locus.positions, int) - locus.start + 1))
    cols = []
    for sample in samples:
        K = int(ploidy_of(sample)) if ploidy_of is not None else 2
        cols.append(":".join(["/".join(["."] * K)] + ["."] * (len(SAMPLE_FIELDS) - 1 + len(fmt_opt))))
    return "\t".join([locus.contig, str(locus.start + 1), locus.name, locus.sequence, ".", ".", "LIMIT", info,
                      ":".join(SAMPLE_FIELDS + tuple(fmt_opt))] + cols)


def _assemble_record_line(locus, samples, per, posteriors, haplotype_posterior_threshold, report=(), ploidy_of=None, encoded=None):
    """The VCF record line of `mchap assemble` from the per-sample summaries (application/assemble.py:144-252,
    baseclass.py:220-302).  per[sample]: genotype [K, M], gprob, sprob, mec, mecp, mci, rcount, rcalls, dp, depth;
    encoded[sample]: the sample's encoded reads (for --report GL)."""
    from .vcfheader import report_fields

    info_opt, fmt_opt = report_fields(report)
    M = len(locus.positions)
    haplotypes, ref_called = call_posterior_haplotypes(posteriors, threshold=haplotype_posterior_threshold)
    labels = {h.tobytes(): i for i, h in enumerate(haplotypes)}
    flt = "PASS"
    if not ref_called:
        labels.pop(haplotypes[0].tobytes())
        if len(haplotypes) == 1:
            flt = "NOA"
    H = len(haplotypes)
    alts = [locus.format_haplotype(h) for h in haplotypes[1:]]
    counts = np.zeros(H, int)
    cols = []
    want_afp = bool({"ACP", "AFP", "AOP", "AOPSUM"} & set(info_opt)) or bool({"ACP", "AFP", "AOP"} & set(fmt_opt))
    acp_sum, aop_sum, aop_not, snvdp_sum = np.zeros(H), np.zeros(H), np.ones(H), np.zeros(M)
    ploidy_total = 0
    for sample, post in zip(samples, posteriors):
        d = per[sample]
        K = len(d["genotype"])
        ploidy_total += K
        a = np.sort([labels.get(np.asarray(h, dtype=np.int8).tobytes(), -1) for h in d["genotype"]])
        a = np.append(a[a >= 0], a[a < 0])
        for x in a:
            if x >= 0:
                counts[x] += 1
        d["alleles"] = a
        fields = ["/".join(str(x) if x >= 0 else "." for x in a), vcfstr(qual_of_prob(d["gprob"])), vcfstr(qual_of_prob(d["sprob"])),
                  vcfstr(float(d["dp"])), str(d["rcount"]), str(d["rcalls"]), str(d["mec"]), vcfstr(float(d["mecp"])),
                  vcfstr(d["gprob"]), vcfstr(d["sprob"]), str(d["mci"])]
        depth = np.asarray(d.get("depth", np.zeros(M)), float)
        if M:
            snvdp_sum += np.round(depth)
        freqs = occur = None
        if want_afp:
            # posterior allele frequencies / occurrences of the record's haplotypes (assemble.py:213-226)
            freqs, occur = np.zeros(H), np.zeros(H)
            hp, fq, oc = post.allele_frequencies()
            index = {np.asarray(h, dtype=np.int8).tobytes(): i for i, h in enumerate(haplotypes)}
            for h, f_, o_ in zip(hp, fq, oc):
                i = index.get(np.asarray(h, dtype=np.int8).tobytes())
                if i is not None:
                    freqs[i], occur[i] = f_, o_
            acp_sum += freqs * K
            aop_sum += occur
            aop_not *= 1 - occur
        for tag in fmt_opt:
            if tag == "AFP":
                fields.append(vcfstr(freqs))
            elif tag == "ACP":
                fields.append(vcfstr(freqs * K))
            elif tag == "AOP":
                fields.append(vcfstr(occur))
            elif tag == "GP":
                fields.append(vcfstr(_genotype_posterior_array(post, labels, K)))
            elif tag == "GL":
                # likelihoods of every genotype over the record's haplotypes (assemble.py:236-244): the exact caller's kernel
                from . import calling

                sr = encoded[sample]
                if M and len(sr["dists"]):
                    llks = calling.genotype_likelihoods(sr["dists"], K, haplotypes, read_counts=sr["counts"])
                else:
                    from math import comb

                    llks = np.zeros(comb(H + K - 1, K), np.float32)
                fields.append(vcfstr(np.asarray(llks, np.float64) / np.log(10)))
            elif tag == "SNVDP":
                fields.append(vcfstr(np.round(depth)) if M else ".")
        cols.append(":".join(fields))
    info = [("AN", int(counts.sum())), ("UAN", int((counts > 0).sum())), ("AC", counts[1:])]
    if not ref_called:
        info.append(("REFMASKED", True))
    info += [("NS", sum(int(np.any(per[s]["alleles"] >= 0)) for s in samples)), ("MCI", sum(int(per[s]["mci"] > 0) for s in samples)),
             ("DP", float(np.nansum([per[s]["dp"] for s in samples])) if M else np.nan),
             ("RCOUNT", int(sum(per[s]["rcount"] for s in samples))), ("END", locus.stop), ("NVAR", M),
             ("SNVPOS", np.array(locus.positions, int) - locus.start + 1)]
    for tag in info_opt:
        if tag == "AFPRIOR":
            info.append(("AFPRIOR", np.full(H, np.nan)))  # (the assembler has no prior allele frequencies)
        elif tag == "ACP":
            info.append(("ACP", acp_sum))
        elif tag == "AFP":
            info.append(("AFP", acp_sum / ploidy_total))
        elif tag == "AOP":
            info.append(("AOP", 1 - aop_not))
        elif tag == "AOPSUM":
            info.append(("AOPSUM", aop_sum))
        elif tag == "SNVDP":
            info.append(("SNVDP", snvdp_sum))
    parts = []
    for k, v in info:
        if isinstance(v, bool):
            if v:
                parts.append(k)
        else:
            parts.append("%s=%s" % (k, vcfstr(v)))
    return "\t".join([locus.contig, str(locus.start + 1), locus.name, locus.sequence, ",".join(alts) if alts else ".", ".", flt,
                      ";".join(parts), ":".join(SAMPLE_FIELDS + tuple(fmt_opt))] + cols)


MAX_SNVS_PER_LOCUS = 126  # what mchap_denovo_fit_batch takes per unit (include/mchap_hip.h MCHAP_MAX_POS)
MAX_ROWS_WIDE = 1024      # distinct read rows of a unit on the general sampler (128-bit haplotype words, ploidy above 8)


def _lib_max_reads():
    from ._lib import MAX_READS

    return MAX_READS


def _row_limit(n_snvs, max_allele, ploidy):
    """Distinct read rows the library takes for a unit of this shape (include/mchap_hip.h): MCHAP_MAX_READS on the fast samplers,
    1024 on the general one (more than 62 SNVs, more than 64 bits of sampled alleles per haplotype, or ploidy above 8)."""
    from ._lib import MAX_READS

    bits = 1 if max_allele <= 2 else (2 if max_allele <= 4 else 3)
    return MAX_ROWS_WIDE if (n_snvs > 62 or n_snvs * bits > 64 or ploidy > 8) else MAX_READS


def _variants_by_contig(variants):
    """{contig: (positions ascending, the records in that order)}: the file order of equal positions is kept (stable sort),
    as DenovoLocus merges the records of a position in the order it is given them."""
    by = {}
    for r in variants:
        by.setdefault(r["chrom"], []).append(r)
    out = {}
    for contig, recs in by.items():
        pos = np.fromiter((r["pos"] for r in recs), dtype=np.int64, count=len(recs))
        o = np.argsort(pos, kind="stable")
        out[contig] = (pos[o], [recs[i] for i in o])
    return out


def _variants_within(by_contig, contig, start, stop):
    """The variant records of a target's interval, by bisection of the contig's sorted positions."""
    if contig not in by_contig:
        return ()
    pos, recs = by_contig[contig]
    lo, hi = np.searchsorted(pos, start + 1, side="left"), np.searchsorted(pos, stop + 1, side="left")  # (pos is 1-based)
    return recs[int(lo):int(hi)]


def assemble_targets(bed_path=None, region=None, region_id=None):
    """The target loci of `mchap assemble`: the lines of a BED4 file, or one --region (application/assemble.py:75-85)."""
    from .io import parse_region

    if bed_path is None and region is None:
        raise ValueError("No region or targets bedfile is specified.")
    if bed_path is not None and region is not None:
        raise ValueError("Cannot combine --targets and --region arguments.")
    if bed_path is not None:
        return read_bed4(bed_path)
    contig, start, stop = parse_region(region)
    return [(contig, start, stop, region_id if region_id is not None else ".")]


def assemble(bed_path, variants_vcf_path, reference_sequences, sample_bams, ploidy=4, inbreeding=None, steps=1000, burn=500,
             chains=2, seed=42, error_rate=0.0024, use_phred=False, haplotype_posterior_threshold=0.20,
             incongruence_threshold=0.60, report=(), temperatures=(1.0,), read_kw=None, targets=None, units_per_block=None,
             shard=None, sampler=None, timings=None, block_path=None, _batch_factory=None, **mcmc_kw):
    """`mchap assemble` over the targets of a BED4 file (or `targets`: a list of (contig, start, stop, name)): yields one
    VCF record line per target (no header).  reference_sequences: {contig: sequence string} or an io.Reference;
    sample_bams: ordered mapping sample name -> BAM path (or pool -> [(sample, path)], or a ReadSource); ploidy /
    inbreeding: a value or {sample: value}; temperatures: a ladder for every sample or {sample: ladder}.

    Batched: the targets are processed in blocks; every (target x sample) unit of a block is encoded and all of them go
    through ONE sampler launch per (ploidy, temperature ladder) present (ragged: loci differ in SNVs, samples in depth),
    the posterior summary (distinct genotypes, SPM, GPM, mode) and the replicate incongruence (MCI) are taken on the
    device, and the block's records are formatted from those few hundred bytes per unit and yielded before the next
    block is read.  A block holds as many units as fit a share of the free HBM (`units_per_block`; a small file is one
    block, one launch).  As in the reference every unit restarts from the same seed (application/baseclass.py:360-388,
    assemble/mcmc.py:140-142), so the result does not depend on the order or the batching of the targets.

    variants_vcf_path may be the list of variant records itself.  sampler: None = the device batch; or a callable
    (units, settings dict) -> per-unit summaries (the oracle-backed replay of the tests).  timings: a dict that receives
    the seconds spent encoding reads, in the sampler (launch to results on the host) and formatting records.

    block_path: None = the host work of a block (read extraction, allele calls, de-duplication, the sampler's descriptors, the
    posterior haplotypes and the record fields) as array operations over the whole block (mchap_amd/blockpath.py, _BlockState
    below) whenever the inputs allow -- alignment files read whole, one file per sample, base qualities ignored --, else locus
    by locus; False = always locus by locus (the definition: tests hold the two against each other line for line); True =
    fail if the block path cannot take the inputs."""
    from .assemble import DenovoMCMC
    from .classes import PosteriorGenotypeDistribution
    from .device import DenovoRaggedBatch, PassesInFlight
    from .io import Reference

    source = _source(sample_bams, error_rate, use_phred, read_kw)
    samples = source.samples
    seed = resolve_seed(seed)
    ploidy_of, inbreeding_of = _per_sample(ploidy, samples), _per_sample(inbreeding, samples)
    temps_of = _per_sample(temperatures if isinstance(temperatures, dict) else tuple(temperatures), samples)
    import time as _time

    variants = variants_vcf_path if isinstance(variants_vcf_path, (list, tuple)) else read_vcf(variants_vcf_path)[1]
    if timings is None:
        timings = {}
    for k_ in ("encode_s", "sampler_s", "format_s"):
        timings.setdefault(k_, 0.0)
    timings.setdefault("units", 0)
    if targets is None:
        targets = read_bed4(bed_path)
    targets = _shard(list(targets), shard)
    fetch = reference_sequences.fetch if isinstance(reference_sequences, Reference) else (lambda c, a, b: reference_sequences[c][a:b])

    def seq_known(contig):
        """False for a reference known by its index only (io.Reference without the FASTA), or a {contig: sequence} mapping whose
        sequence object says so itself (attribute `known`: the tests' stand-in for such a reference)."""
        if isinstance(reference_sequences, Reference):
            return reference_sequences.known
        return bool(getattr(reference_sequences[contig], "known", True))
    by_contig = _variants_by_contig(variants)
    if units_per_block is None:
        # device bytes of one unit: its traces (chains x steps x (ploidy words + llk)), the floor of the per-chain likelihood
        # cache (1024 entries of 16 bytes) and the tables of the workspace, a typical read tensor; two blocks are resident at a
        # time (below), each with up to four sampler passes whose caches and decision contexts grow into fixed budgets of 4 + 6
        # GiB (csrc/mchap_hip.hip CACHE_BUDGET, CTX_BUDGET; each at most an eighth of what is free): set aside.  A block is at most
        # 32 768 units: large enough that its launch is set by the average chain and not by its slowest, small enough that
        # a big job is several blocks whose host and device work overlap
        kmax = max(int(ploidy_of(s_)) for s_ in samples)
        units_per_block = min(32768, device_unit_budget(chains * steps * (kmax + 1) * 8 + chains * 1024 * 16 + (256 << 10), fraction=0.2,
                                                             reserve=2 * 4 * (10 << 30)))
    loci_per_block = max(1, units_per_block // max(1, len(samples)))

    def start_block(block, stream):
        """Stage 1 of a block: loci, read encoding (host), the sampler launches enqueued on `stream` (not waited for)."""
        loci = [DenovoLocus(contig, start, stop, name, _variants_within(by_contig, contig, start, stop), fetch(contig, start, stop),
                            sequence_known=seq_known(contig))
                for contig, start, stop, name in block]
        encoded = {}
        units, where = [], []
        t0_ = _time.perf_counter()
        skipped = {}
        for li, locus in enumerate(loci):
            M = len(locus.positions)
            if M > MAX_SNVS_PER_LOCUS:
                # beyond what the library takes (the reference has no limit): the target is left out of the output with a
                # warning instead of ending a run of many targets
                skipped[li] = "%d SNVs (at most %d per target)" % (M, MAX_SNVS_PER_LOCUS)
                continue
            mine = []
            for sample in samples:
                sr = source.reads(locus, sample)
                encoded[(li, sample)] = sr
                if M == 0:
                    continue  # nothing to sample: the empty genotype with probability 1
                dists, counts = sr["dists"], sr["counts"]
                if len(dists) == 0:  # no reads: one all-gap read (assemble/mcmc.py:132-137)
                    dists, counts = np.full((1, M, int(max(locus.n_alleles))), np.nan), None
                lim = _row_limit(M, int(max(locus.n_alleles)), int(ploidy_of(sample)))
                if len(dists) > lim:  # (a deep target of a wide shape: its record says LIMIT, the other targets go on)
                    skipped[li] = "sample %s: %d distinct read rows (at most %d for a target of this shape)" % (sample, len(dists), lim)
                    break
                mine.append((dict(reads=dists, counts=counts, n_alleles=locus.n_alleles, ploidy=int(ploidy_of(sample)),
                                  inbreeding=inbreeding_of(sample), stream_id=0, temps=tuple(temps_of(sample))), (li, sample)))
            if li not in skipped:
                for u_, w_ in mine:
                    units.append(u_)
                    where.append(w_)
        t1_ = _time.perf_counter()
        timings["encode_s"] += t1_ - t0_
        timings["units"] += len(units)
        state = dict(loci=loci, encoded=encoded, units=units, where=where, summaries={}, pending=[], stream=stream, skipped=skipped)
        if units and sampler is not None:
            settings = dict(steps=steps, chains=chains, seed=seed, burn=burn, incongruence_threshold=incongruence_threshold, **mcmc_kw)
            for w_, res in zip(where, sampler(units, settings)):
                state["summaries"][w_] = res
        elif units:
            import torch

            # one launch per (ploidy, temperature ladder) present (usually one): the library's fast samplers take one ploidy
            # per launch (mixed ploidies would run on the general lanes-over-chains kernel), and a ladder is a launch setting
            # ... and units wider than the fast samplers take (more than 62 SNVs or 64 bits of alleles per haplotype) go to a launch
            # of their own: they run on the general sampler with 128-bit haplotype words, which must not slow the others down
            groups = {}
            for i, u in enumerate(units):
                M_, A_ = u["reads"].shape[1], u["reads"].shape[2]
                wide = M_ > 62 or M_ * (1 if A_ <= 2 else 2 if A_ <= 4 else 3) > 64
                groups.setdefault((u["ploidy"], u["temps"], wide), []).append(i)
            with torch.cuda.stream(stream) if stream is not None else _nullcontext():
                # (several groups: their launches go out on separate HIP streams and fill each other's thin phases)
                flight = PassesInFlight(min(4, len(groups))) if len(groups) > 1 else None
                for (K, temps, _wide), idx in groups.items():
                    model = DenovoMCMC(ploidy=K, n_alleles=[2], inbreeding=None, steps=steps, chains=chains, random_seed=seed,
                                       temperatures=temps, **mcmc_kw)
                    batch = DenovoRaggedBatch(model, [units[i] for i in idx])
                    if flight is not None:
                        flight.submit(lambda b=batch: b.run(burn, incongruence_threshold=incongruence_threshold))
                    else:
                        batch.run(burn, incongruence_threshold=incongruence_threshold)
                    state["pending"].append((idx, batch))
                if flight is not None:
                    flight.join()
        timings["sampler_s"] += _time.perf_counter() - t1_
        return state

    def finish_block(state):
        """Stage 2: the block's results on the host (waits for its launches only), its records formatted and yielded."""
        t1_ = _time.perf_counter()
        loci, encoded, where, summaries = state["loci"], state["encoded"], state["where"], state["summaries"]
        if state["pending"]:
            import torch

            with torch.cuda.stream(state["stream"]) if state["stream"] is not None else _nullcontext():
                for idx, batch in state["pending"]:
                    for i, res in zip(idx, batch.results(raise_on_limit=False)):
                        summaries[where[i]] = res
                        if res.get("limit"):
                            state["skipped"].setdefault(where[i][0], "sample %s: %s" % (where[i][1], res["limit"]))
            state["pending"] = []  # (the block's device buffers go back to the allocator)
        t2_ = _time.perf_counter()
        timings["sampler_s"] += t2_ - t1_
        for li, locus in enumerate(loci):
            M = len(locus.positions)
            if li in state["skipped"]:
                import sys

                sys.stderr.write("mchap_amd assemble: target %s (%s:%d-%d) not assembled (record written with FILTER=LIMIT): %s\n" % (
                    locus.name, locus.contig, locus.start + 1, locus.stop, state["skipped"][li]))
                timings["limit_records"] = timings.get("limit_records", 0) + 1
                yield _limit_record_line(locus, samples, report, ploidy_of)
                continue
            per, posteriors = {}, []
            for sample in samples:
                sr = encoded[(li, sample)]
                K = int(ploidy_of(sample))
                if M == 0:
                    res = dict(genotypes=np.zeros((1, K, 0), np.int8), probabilities=np.ones(1), spm=1.0, gpm=1.0,
                               mode_genotype=np.zeros((K, 0), np.int8), mci=0)
                else:
                    res = summaries[(li, sample)]
                posteriors.append(PosteriorGenotypeDistribution(res["genotypes"], res["probabilities"]))
                calls, depth = sr["calls"], sr["depth"]
                mec = _mec(calls, res["mode_genotype"])
                denom = int((calls >= 0).sum())
                per[sample] = dict(genotype=res["mode_genotype"], gprob=float(res["gpm"]), sprob=float(res["spm"]), mec=mec,
                                   mecp=mec / denom if denom > 0 else np.nan, mci=int(res["mci"]), rcount=len(calls), rcalls=denom,
                                   dp=np.round(np.mean(depth)) if len(depth) else np.nan, depth=depth)
            line = _assemble_record_line(locus, samples, per, posteriors, haplotype_posterior_threshold, report, ploidy_of,
                                         {s_: encoded[(li, s_)] for s_ in samples})
            timings["format_s"] += _time.perf_counter() - t2_
            yield line
            t2_ = _time.perf_counter()

    # Two blocks in flight: while the device samples block i the host encodes block i + 1, and while it samples block i + 1
    # the host formats block i.  Each block's launches go to one of two side streams (a block's results are read on its own
    # stream, so fetching them does not wait for the next block's launches); a job of one block runs on the current stream.
    blocks = list(_blocks(targets, loci_per_block))
    streams = [None, None]
    if len(blocks) > 1 and sampler is None and _batch_factory is None:
        import torch

        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    fast = block_path is not False and sampler is None and _block_path_takes(source)
    if block_path is True and not fast:
        raise ValueError("block_path=True: the inputs need the per-locus path")

    def start(block, stream):
        if fast:
            from .blockpath import BlockPathUnavailable

            try:
                return _BlockState(block, stream, locals_of_run)
            except BlockPathUnavailable:
                if block_path is True:
                    raise
        return start_block(block, stream)

    def finish(state):
        if isinstance(state, _BlockState):
            return state.finish()
        return finish_block(state)

    locals_of_run = dict(source=source, samples=samples, ploidy_of=ploidy_of, inbreeding_of=inbreeding_of, temps_of=temps_of,
                         by_contig=by_contig, fetch=fetch, seq_known=seq_known, timings=timings, steps=steps, chains=chains, seed=seed,
                         burn=burn, incongruence_threshold=incongruence_threshold, mcmc_kw=mcmc_kw, report=report,
                         threshold=haplotype_posterior_threshold, batch_factory=_batch_factory)
    prev = None
    for bi, block in enumerate(blocks):
        cur = start(block, streams[bi % 2])
        if prev is not None:
            yield from finish(prev)
        prev = cur
    if prev is not None:
        yield from finish(prev)


def _block_path_takes(source):
    """The array path of a block reads every sample from one alignment file held whole in memory (or from pileup matrices the
    caller has already), base qualities ignored."""
    from .io import BamFile

    if isinstance(source, MatrixSource):
        return not source.use_phred
    if not isinstance(source, ReadSource) or source.use_phred:
        return False
    for pairs in source.pools.values():
        if len(pairs) != 1:
            return False
        bam = source.bams[pairs[0][1]]
        if not isinstance(bam, BamFile) or (bam.index is not None and len(bam.data) > source.WHOLE_FILE_BYTES):
            return False
    return True


def _rounded_text(r):
    """io._number_text of a value already rounded by np.round (an array's worth at a time)."""
    if r != r:
        return "."
    if abs(r) != float("inf") and r == int(r):
        return str(int(r))
    return repr(r)[:16]


class _BlockState:
    """One block of targets of `assemble` on the array path: __init__ is stage 1 (loci, extraction and encoding of every
    sample over all loci at once, the sampler launches enqueued), finish() stage 2 (summaries as arrays, records)."""

    def __init__(self, block, stream, run):
        import time as _time

        from . import blockpath as bp
        from .assemble import DenovoMCMC
        from .device import DenovoRaggedBatch, PassesInFlight

        self.run, self.stream = run, stream
        source, samples, timings = run["source"], run["samples"], run["timings"]
        self.loci = loci = [DenovoLocus(c, a, b, nm, _variants_within(run["by_contig"], c, a, b), run["fetch"](c, a, b),
                                        sequence_known=run["seq_known"](c)) for c, a, b, nm in block]
        t0 = _time.perf_counter()
        self.skipped = {li: "%d SNVs (at most %d per target)" % (len(l.positions), MAX_SNVS_PER_LOCUS)
                        for li, l in enumerate(loci) if len(l.positions) > MAX_SNVS_PER_LOCUS}
        self.xi_of = {}
        self.lx = lx = [l for li, l in enumerate(loci) if li not in self.skipped]
        for li in range(len(loci)):
            if li not in self.skipped:
                self.xi_of[li] = len(self.xi_of)
        tables = bp.locus_tables(lx)
        M, snv_start, _ = tables
        self.M = M
        self.enc = []
        lut = bp.allele_lut(lx, int(snv_start[-1]))
        if isinstance(source, ReadSource) and source.workers > 1:
            # files not parsed yet: inflated and walked side by side (zlib and the library's record walk release the interpreter lock)
            todo = [b for b in dict.fromkeys(source.bams[source.pools[s_][0][1]] for s_ in samples) if getattr(b, "_all", 0) is None]
            if len(todo) > 1:
                from concurrent.futures import ThreadPoolExecutor

                tp = _time.perf_counter()
                with ThreadPoolExecutor(max_workers=min(source.workers, len(todo))) as ex:
                    list(ex.map(lambda b: b.columns(), todo))
                timings["bam_parse_s"] = timings.get("bam_parse_s", 0.0) + _time.perf_counter() - tp
        for sample in samples:
            if isinstance(source, MatrixSource):
                self.enc.append(bp.encode_block(lx, bp.pile_from_matrices(lx, [source.codes(l.name, sample) for l in lx], tables=tables), lut))
                continue
            name, path = source.pools[sample][0]
            tp = _time.perf_counter()
            cols = source.bams[path].columns()   # (the first use of a file inflates and parses it)
            timings["bam_parse_s"] = timings.get("bam_parse_s", 0.0) + _time.perf_counter() - tp
            self.enc.append(bp.encode_block(lx, bp.extract_block(lx, cols, name, tables=tables, **source.filter), lut))
        # the sampler's units: every (sample, locus with SNVs), one launch per (ploidy, temperature ladder, wide) present
        nal = np.fromiter((a for l in lx for a in l.n_alleles), dtype=np.int8, count=int(snv_start[-1]))
        amax = np.array([max(l.n_alleles) if l.n_alleles else 0 for l in lx], dtype=np.int64)
        bits = np.where(amax <= 2, 1, np.where(amax <= 4, 2, 3))
        wide = (M > 62) | (M * bits > 64)
        sampled = M > 0
        # a unit with more distinct read rows than the library takes for its shape: the target's record says LIMIT (the batch
        # constructor would refuse the whole launch -- and with it every other target of the block)
        li_of = {xi: li for li, xi in self.xi_of.items()}
        for si, sample in enumerate(samples):
            nu = self.enc[si].urow_start[1:] - self.enc[si].urow_start[:-1]
            K_ = int(run["ploidy_of"](sample))
            lim = np.where(wide | (K_ > 8), MAX_ROWS_WIDE, _lib_max_reads())
            for xi in np.flatnonzero(sampled & (nu > lim)):
                self.skipped.setdefault(li_of[int(xi)], "sample %s: %d distinct read rows (at most %d for a target of this shape)" % (
                    sample, int(nu[xi]), int(lim[xi])))
                sampled[xi] = False
        self.batches = []   # (batch, samples [n], loci [m], unit arrays): unit = position of the sample * m + position of the locus
        self.unit_at = np.full((len(samples), len(lx), 2), -1, dtype=np.int64)  # (batch, unit) of every (sample, locus)
        groups = {}
        for si, sample in enumerate(samples):
            groups.setdefault((int(run["ploidy_of"](sample)), tuple(run["temps_of"](sample))), []).append(si)
        plans = []
        for (K, temps), sis in groups.items():
            for w in (False, True):
                use = np.flatnonzero(sampled & (wide == w))
                if len(use):
                    plans.append((K, temps, sis, use))
        t1 = _time.perf_counter()
        timings["encode_s"] += t1 - t0
        factory = run["batch_factory"] or DenovoRaggedBatch.from_calls
        if run["batch_factory"] is None:
            import torch

            ctx = torch.cuda.stream(stream) if stream is not None else _nullcontext()
        else:
            ctx = _nullcontext()
        with ctx:
            flight = PassesInFlight(min(4, len(plans))) if (len(plans) > 1 and run["batch_factory"] is None) else None
            for bi, (K, temps, sis, use) in enumerate(plans):
                parts = [bp.unit_inputs(self.enc[si], use) for si in sis]
                m = len(use)
                c_base = np.cumsum([0] + [len(p[0]) for p in parts])
                n_base = np.cumsum([0] + [len(p[1]) for p in parts])
                u = dict(calls=np.concatenate([p[0] for p in parts]), counts=np.concatenate([p[1] for p in parts]),
                         n_reads=np.concatenate([p[2] for p in parts]),
                         reads_off=np.concatenate([p[3] + c_base[i] for i, p in enumerate(parts)]),
                         counts_off=np.concatenate([np.where(p[4] >= 0, p[4] + n_base[i], -1) for i, p in enumerate(parts)]),
                         n_pos=np.tile(M[use], len(sis)), max_allele=np.tile(amax[use], len(sis)), nalleles_off=np.tile(snv_start[use], len(sis)),
                         ploidy=np.full(m * len(sis), K, dtype=np.int64))
                F = np.array([np.nan if run["inbreeding_of"](samples[si]) is None else float(run["inbreeding_of"](samples[si])) for si in sis])
                model = DenovoMCMC(ploidy=K, n_alleles=[2], inbreeding=None, steps=run["steps"], chains=run["chains"], random_seed=run["seed"],
                                   temperatures=temps, **run["mcmc_kw"])
                batch = factory(model, u["calls"], u["reads_off"], u["n_reads"], u["n_pos"], u["max_allele"], u["ploidy"], u["counts"],
                                u["counts_off"], nal, u["nalleles_off"], inbreeding=np.repeat(F, m), error_rate=source.error_rate)
                go = (lambda b=batch: b.run(run["burn"], incongruence_threshold=run["incongruence_threshold"]))
                if flight is not None:
                    flight.submit(go)
                else:
                    go()
                for k, si in enumerate(sis):
                    self.unit_at[si, use, 0] = bi
                    self.unit_at[si, use, 1] = k * m + np.arange(m)
                u["bits"] = np.tile(bits[use], len(sis))
                self.batches.append((batch, u, K))
            if flight is not None:
                flight.join()
        timings["units"] += int(sum(len(b[1]["n_reads"]) for b in self.batches))
        timings["sampler_s"] += _time.perf_counter() - t1
        timings["launch_s"] = timings.get("launch_s", 0.0) + _time.perf_counter() - t1   # (part of sampler_s: descriptors, H2D, enqueue)

    # ---- stage 2 ----
    def _summaries(self, batch, u, K):
        """Per unit of a batch, from the device's summary arrays: GQ / SQ / GPM / SPM / MCI / MEC values, the listed haplotypes
        (allele rows as bytes) with their expected dosages, the mode genotype's haplotypes; None for the units whose summary
        is not complete in the arrays (their loci go through the per-locus formatter)."""
        from . import blockpath as bp

        run = self.run
        arr = batch.summary_arrays()
        plain = arr["plain"]
        if not plain.any():
            return None
        U = len(plain)
        M, total = u["n_pos"], arr["total"]
        n = np.where(plain, arr["n"], 0).astype(np.int64)
        gu, gi = bp._ragged_arange(n)
        wide = getattr(batch, "wph", 1) != 1
        if wide:
            # a batch of the general sampler (round 5): a haplotype is two words -- the distinct pairs of the batch are numbered,
            # everything below compares and groups the numbers, and unpack_words gets the pairs back
            pu_ = np.flatnonzero(plain)
            pairs = np.concatenate([arr["words"][:, :K].reshape(-1, 2), arr["mode_words"][pu_, :K].reshape(-1, 2)])
            upairs, inv = np.unique(pairs, axis=0, return_inverse=True)
            inv = inv.reshape(-1).astype(np.uint64)
            W = inv[: len(arr["words"]) * K].reshape(len(arr["words"]), K)
            mode_ids = inv[len(arr["words"]) * K:].reshape(len(pu_), K)
        else:
            W = arr["words"][:, :K]
        p = arr["counts"] / total
        # allele_frequencies(dosage=True) of every unit (classes.py): one term per (genotype, distinct haplotype), summed in
        # genotype order; haplotypes in order of first appearance
        same = W[:, :, None] == W[:, None, :]
        dose = same.sum(axis=2)
        first = ~(same & np.tril(np.ones((K, K), dtype=bool), -1)[None]).any(axis=2)
        ent_u = np.broadcast_to(gu[:, None], W.shape)[first]
        ent_w, ent_p, ent_d = W[first], np.broadcast_to(p[:, None], W.shape)[first], dose[first]
        o = np.lexsort((ent_w, ent_u))
        su, sw = ent_u[o], ent_w[o]
        new = np.r_[True, (su[1:] != su[:-1]) | (sw[1:] != sw[:-1])] if len(o) else np.zeros(0, dtype=bool)
        first_idx = o[np.flatnonzero(new)]
        order = np.argsort(first_idx, kind="stable")
        rank = np.empty(len(order), dtype=np.int64)
        rank[order] = np.arange(len(order))
        gid = np.empty(len(o), dtype=np.int64)
        gid[o] = rank[np.cumsum(new) - 1]
        H = len(first_idx)
        weight = np.bincount(gid, weights=ent_p * ent_d, minlength=H)
        occur = np.bincount(gid, weights=ent_p, minlength=H)
        hap_u, hap_w = ent_u[first_idx[order]], ent_w[first_idx[order]]
        listed = occur >= run["threshold"]
        lu, lw, lwt = hap_u[listed], hap_w[listed], weight[listed]
        pu = np.flatnonzero(plain)
        if wide:
            words = upairs[np.concatenate([lw, mode_ids.reshape(-1)]).astype(np.int64)]   # [n, 2]
        else:
            words = np.concatenate([lw, arr["mode_words"][pu, :K].reshape(-1)])
        unit = np.concatenate([lu, np.repeat(pu, K)])
        fixed_off = batch.units_host["fixed_off"].astype(np.int64)
        flat, cell_of = bp.unpack_words(words, unit, arr["fixed"], fixed_off, M, u["bits"])
        B = flat.tobytes()
        lstart = np.zeros(U + 1, dtype=np.int64)
        np.cumsum(np.bincount(lu, minlength=U), out=lstart[1:])
        mode_slot = np.full(U, -1, dtype=np.int64)
        mode_slot[pu] = len(lw) + np.arange(len(pu)) * K      # index of the unit's first mode haplotype among `words`
        # MEC of the mode genotype over the unit's reads: distinct rows x their counts (encoding/integer/stats.py:18-39)
        R = u["n_reads"]
        ru, rk = bp._ragged_arange(R)
        cr, cj = bp._ragged_arange(M[ru])
        cu = ru[cr]
        c = u["calls"][u["reads_off"][cu] + rk[cr] * M[cu] + cj]
        row_first = np.cumsum(M[ru]) - M[ru]
        best = None
        ms_ = np.where(mode_slot >= 0, mode_slot, 0)
        for h in range(K):
            g = flat[cell_of[ms_[cu] + h] + cj]
            per_row = np.add.reduceat(((c != g) & (c >= 0)).astype(np.int64), row_first)
            best = per_row if best is None else np.minimum(best, per_row)
        cnt = np.where(u["counts_off"][ru] >= 0, u["counts"][np.maximum(u["counts_off"][ru], 0) + rk] if len(u["counts"]) else 1, 1)
        mec = np.bincount(ru, weights=best * cnt, minlength=U).astype(np.int64)
        gpm, spm = arr["stats"][:, 1], arr["stats"][:, 0]
        unit6 = 10 ** 6

        def qual(prob):  # io.qual_of_prob over an array
            kept = np.floor(np.minimum(prob, 1 - 0.1 ** 6) * unit6) / unit6
            with np.errstate(invalid="ignore"):
                return np.round(-10 * np.log10(1 - kept))
        return dict(plain=plain, B=B, cell_of=cell_of, lstart=lstart, lweight=lwt.tolist(), mode_slot=mode_slot, mec=mec,
                    gq=qual(gpm), sq=qual(spm), gpm=np.round(gpm, 3).tolist(), spm=np.round(spm, 3).tolist(), mci=arr["mci"], M=M)

    def finish(self):
        import sys
        import time as _time

        run = self.run
        samples, timings, report = run["samples"], run["timings"], run["report"]
        t1 = _time.perf_counter()
        if run["batch_factory"] is None:
            import torch

            ctx = torch.cuda.stream(self.stream) if self.stream is not None else _nullcontext()
        else:
            ctx = _nullcontext()
        with ctx:
            if run["batch_factory"] is None:
                torch.cuda.current_stream().synchronize()
                timings["device_wait_s"] = timings.get("device_wait_s", 0.0) + _time.perf_counter() - t1   # (part of sampler_s)
            sums = [self._summaries(*b) for b in self.batches]
            # the loci that go through the per-locus formatter (no SNVs, --report fields, a unit whose summary is not in the arrays)
            # and the units it will ask for
            slow_x = (self.M == 0) | bool(report)
            for si in range(len(samples)):
                bi_, uu = self.unit_at[si, :, 0], self.unit_at[si, :, 1]
                for b, sm in enumerate(sums):
                    m = bi_ == b
                    slow_x[m] |= True if sm is None else ~sm["plain"][uu[m]]
            self.slow_res = []
            for b, (batch, u, K) in enumerate(self.batches):
                need = sorted(int(x) for si in range(len(samples)) for x in self.unit_at[si, (self.unit_at[si, :, 0] == b) & slow_x, 1])
                self.slow_res.append(dict(zip(need, batch.results(raise_on_limit=False, only=need))) if need else {})
            self.slow_x = slow_x
            # MEC / MECP of every (sample, locus) whose unit is in the arrays
            self.mec = np.zeros((len(samples), len(self.lx)), dtype=np.int64)
            for si in range(len(samples)):
                bi_, uu = self.unit_at[si, :, 0], self.unit_at[si, :, 1]
                for b, sm in enumerate(sums):
                    m = (bi_ == b) & ~slow_x
                    if sm is not None and m.any():
                        self.mec[si, m] = sm["mec"][uu[m]]
            with np.errstate(invalid="ignore", divide="ignore"):
                self.mecp = [np.round(self.mec[si] / self.enc[si].rcalls, 3).tolist() for si in range(len(samples))]
        t2 = _time.perf_counter()
        timings["sampler_s"] += t2 - t1
        S = len(samples)
        K_of = [int(run["ploidy_of"](s_)) for s_ in samples]
        fmt_head = ":".join(SAMPLE_FIELDS)
        for li, locus in enumerate(self.loci):
            if li in self.skipped:
                sys.stderr.write("mchap_amd assemble: target %s (%s:%d-%d) not assembled (record written with FILTER=LIMIT): %s\n" % (
                    locus.name, locus.contig, locus.start + 1, locus.stop, self.skipped[li]))
                timings["limit_records"] = timings.get("limit_records", 0) + 1
                yield _limit_record_line(locus, samples, report, run["ploidy_of"])
                continue
            xi = self.xi_of[li]
            M = int(self.M[xi])
            at = self.unit_at[:, xi]
            ok = not self.slow_x[xi]
            if not ok:
                line = self._slow_record(li, xi, sums)
                if line is None:   # (a unit beyond a limit)
                    line = _limit_record_line(locus, samples, report, run["ploidy_of"])
                timings["format_s"] += _time.perf_counter() - t2
                yield line
                t2 = _time.perf_counter()
                continue
            # call_posterior_haplotypes: the haplotypes any sample lists, ranked by expected dosage summed over the samples
            index, score = {}, []
            for si in range(S):
                sm = sums[at[si, 0]]
                u = int(at[si, 1])
                B, cell_of, lw = sm["B"], sm["cell_of"], sm["lweight"]
                for k in range(int(sm["lstart"][u]), int(sm["lstart"][u + 1])):
                    c0 = int(cell_of[k])
                    key = B[c0:c0 + M]
                    i = index.get(key)
                    if i is None:
                        index[key] = len(score)
                        score.append(0.0 + lw[k])
                    else:
                        score[i] += lw[k]
            ref = bytes(M)
            ref_called = ref in index
            keys = [k_ for k_ in index if k_ != ref]
            vals = [score[index[k_]] for k_ in keys]
            keys.append(ref)
            vals.append((max(vals) if vals else -1.0) + 1.0)
            if len(keys) > 2:
                if len(set(vals)) == len(vals):   # (no ties: the descending order is unique)
                    keys = [keys[i] for i in sorted(range(len(vals)), key=vals.__getitem__, reverse=True)]
                else:
                    keys = [keys[i] for i in np.flip(np.argsort(np.array(vals))).tolist()]
            elif len(keys) == 2:
                keys = [keys[1], keys[0]]
            labels = {k_: i for i, k_ in enumerate(keys)}
            flt = "PASS"
            if not ref_called:
                labels.pop(ref)
                if len(keys) == 1:
                    flt = "NOA"
            H = len(keys)
            seq0 = list(locus.sequence)
            alts = []
            rel = [p_ - locus.start for p_ in locus.positions]
            for key in keys[1:]:
                chars = seq0[:]
                for r_, tup, a in zip(rel, locus.alleles, key):
                    chars[r_] = tup[a]
                alts.append("".join(chars))
            counts = [0] * H
            cols, ns, n_mci, dp_sum, rc_sum = [], 0, 0, 0.0, 0
            any_dp = False
            for si in range(S):
                sm = sums[at[si, 0]]
                u = int(at[si, 1])
                K = K_of[si]
                enc = self.enc[si]
                m0 = int(sm["mode_slot"][u])
                cell_of, B = sm["cell_of"], sm["B"]
                a = []
                for h in range(K):
                    c0 = int(cell_of[m0 + h])
                    a.append(labels.get(B[c0:c0 + M], -1))
                a.sort()
                neg = [x for x in a if x < 0]
                a = [x for x in a if x >= 0]
                for x in a:
                    counts[x] += 1
                if a:
                    ns += 1
                mci = int(sm["mci"][u])
                n_mci += mci > 0
                mec = int(self.mec[si, xi])
                rcalls, rcount, dp = int(enc.rcalls[xi]), int(enc.rcount[xi]), float(enc.dp[xi])
                if dp == dp:
                    dp_sum += dp
                rc_sum += rcount
                mecp = _rounded_text(self.mecp[si][xi]) if rcalls > 0 else "."
                cols.append(":".join(["/".join([str(x) for x in a] + ["."] * len(neg)), str(int(sm["gq"][u])), str(int(sm["sq"][u])),
                                      _rounded_text(dp), str(rcount), str(rcalls), str(mec), mecp,
                                      _rounded_text(sm["gpm"][u]), _rounded_text(sm["spm"][u]), str(mci)]))
            info = ["AN=%d" % sum(counts), "UAN=%d" % sum(1 for x in counts if x > 0), "AC=" + (",".join(str(x) for x in counts[1:]) if H > 1 else ".")]
            if not ref_called:
                info.append("REFMASKED")
            info += ["NS=%d" % ns, "MCI=%d" % n_mci, "DP=" + _rounded_text(dp_sum), "RCOUNT=%d" % rc_sum, "END=%d" % locus.stop,
                     "NVAR=%d" % M, "SNVPOS=" + ",".join(str(r_ + 1) for r_ in rel)]
            line = "\t".join([locus.contig, str(locus.start + 1), locus.name, locus.sequence, ",".join(alts) if alts else ".", ".", flt,
                              ";".join(info), fmt_head] + cols)
            timings["format_s"] += _time.perf_counter() - t2
            yield line
            t2 = _time.perf_counter()

    def _slow_record(self, li, xi, sums):
        """A locus the array formatter does not take (no SNVs, --report fields, a unit whose summary is not in the arrays): the
        per-locus formatter on the same data.  None: a unit of the locus is beyond a limit of the library."""
        import sys

        from .classes import PosteriorGenotypeDistribution

        run = self.run
        samples, report = run["samples"], run["report"]
        locus = self.loci[li]
        M = int(self.M[xi])
        per, posteriors, encoded = {}, [], {}
        for si, sample in enumerate(samples):
            K = int(run["ploidy_of"](sample))
            d = self.enc[si].per_locus(xi)
            calls, depth = d["calls"], d["depth"]
            sr = dict(calls=calls, depth=depth, counts=d["counts"])
            if "GL" in report and M:
                sr["dists"] = encoding.encode_read_distributions(locus.n_alleles, d["ucalls"], None, error_rate=run["source"].error_rate)
            encoded[sample] = sr
            if M == 0:
                res = dict(genotypes=np.zeros((1, K, 0), np.int8), probabilities=np.ones(1), spm=1.0, gpm=1.0,
                           mode_genotype=np.zeros((K, 0), np.int8), mci=0)
            else:
                bi, u = (int(x) for x in self.unit_at[si, xi])
                res = self.slow_res[bi][u]
                if res.get("limit"):
                    sys.stderr.write("mchap_amd assemble: target %s (%s:%d-%d) not assembled (record written with FILTER=LIMIT): sample %s: %s\n" % (
                        locus.name, locus.contig, locus.start + 1, locus.stop, sample, res["limit"]))
                    run["timings"]["limit_records"] = run["timings"].get("limit_records", 0) + 1
                    return None
            posteriors.append(PosteriorGenotypeDistribution(res["genotypes"], res["probabilities"]))
            mec = _mec(calls, res["mode_genotype"])
            denom = int((calls >= 0).sum())
            per[sample] = dict(genotype=res["mode_genotype"], gprob=float(res["gpm"]), sprob=float(res["spm"]), mec=mec,
                               mecp=mec / denom if denom > 0 else np.nan, mci=int(res["mci"]), rcount=len(calls), rcalls=denom,
                               dp=np.round(np.mean(depth)) if len(depth) else np.nan, depth=depth)
        return _assemble_record_line(locus, samples, per, posteriors, run["threshold"], report, run["ploidy_of"], encoded)


# ---------------------------------------------------------------------------------------------------------
# mchap call (application/call.py:52-199): Gibbs sampler over the known haplotypes of an input VCF
# ---------------------------------------------------------------------------------------------------------
def call(vcf_path, sample_bams, ploidy=2, report=(), base_error_rate=0.0024, use_base_phred_scores=False,
         prior_frequencies_tag=None, inbreeding=None, steps=2000, burn=1000, chains=2, seed=42,
         incongruence_threshold=0.60, step_type="Gibbs", filter_input_haplotypes=None, read_kw=None, records_per_block=1024,
         shard=None):
    """`mchap call`: yields one VCF record line per record of the input VCF (no header).  The records are processed in
    blocks; the (record x sample) units of a block are grouped by shape and each group is one launch of the sampler kernel
    (CallingMCMC.fit_batch), cut into several when its traces and per-chain likelihood tables would not fit the free HBM.
    As in the reference every unit restarts from the same seed."""
    from .calling_mcmc import CallingMCMC, CallSummary, GenotypeAllelesMultiTrace
    from . import _lib, calling

    source = _source(sample_bams, base_error_rate, use_base_phred_scores, read_kw)
    samples = source.samples
    seed = resolve_seed(seed)
    _, records = read_vcf(vcf_path)
    records = _shard(records, shard)
    report = tuple(report)
    ploidy_of, inbreeding_of = _per_sample(ploidy, samples), _per_sample(inbreeding, samples)
    allele_filter = _allele_filter(vcf_path, filter_input_haplotypes)
    for block in _blocks(records, records_per_block):
        units = _exact_units(block, source, samples, prior_frequencies_tag, allele_filter)
        # haplotypes with zero prior frequency (and a masked reference) are left out of the sampler (call.py:72-84)
        groups = {}
        for ri, unit in enumerate(units):
            locus = unit["locus"]
            H, M = locus.haplotypes.shape
            keep = np.ones(H, bool)
            if locus.mask_reference_allele:
                keep[0] = False
            keep &= ~(np.nan_to_num(locus.frequencies, nan=1.0) == 0)
            unit["keep"] = keep
            if unit["invalid"] is None and keep.sum() == 0:
                unit["invalid"] = "NOA"
            if unit["invalid"] is not None or M == 0:
                continue
            for s in samples:
                groups.setdefault((M, int(max(locus.n_alleles)), int(keep.sum()), int(ploidy_of(s))), []).append((ri, s))
        results = {}
        pending = []
        for (M, A, H, K), all_members in groups.items():
            Rmax = max(max(len(units[ri]["reads"][s]["dists"]), 1) for ri, s in all_members)
            # device bytes per unit: reads, traces, and the chains' tables of remembered likelihoods (the workspace)
            ws1 = int(_lib.lib().mchap_call_mcmc_workspace_bytes_for(1, Rmax, H, K, int(steps), int(chains)))
            per_unit = Rmax * M * A * 8 + Rmax * 8 + chains * steps * (K + 1) * 8 + ws1 + 4096
            for members in _blocks(all_members, device_unit_budget(per_unit)):
                U = len(members)
                reads = np.full((U, Rmax, M, A), np.nan)
                counts = np.zeros((U, Rmax), dtype=np.int64)
                haps = np.zeros((U, H, M), dtype=np.int8)
                has_prior = inbreeding_of(members[0][1]) is not None
                Fs, frs = np.zeros(U), np.zeros((U, H))
                for i, (ri, s) in enumerate(members):
                    sr, locus = units[ri]["reads"][s], units[ri]["locus"]
                    n = len(sr["dists"])
                    if n:
                        reads[i, :n] = sr["dists"]
                        counts[i, :n] = sr["counts"]
                    haps[i] = locus.haplotypes[units[ri]["keep"]]
                    if has_prior:
                        Fs[i] = inbreeding_of(s)
                        frs[i] = locus.frequencies[units[ri]["keep"]]
                model = CallingMCMC(ploidy=K, haplotypes=haps[0], steps=steps, chains=chains, random_seed=seed, step_type=step_type)
                try:
                    # (the traces stay on the device: what a record reads off them is summarised there; the shapes of a block of
                    # records are enqueued on streams of their own and waited for together -- round 5)
                    handle = model.start_batch_summaries(reads, counts, haplotypes=haps, prior=(Fs, frs) if has_prior else None,
                                                         stream_ids=np.zeros(U, dtype=np.uint64), burn=burn,
                                                         incongruence_threshold=incongruence_threshold, stream=_call_stream(len(pending)))
                except NotImplementedError as e:  # (a shape beyond the sampler's limits: FILTER=LIMIT records, the file goes on)
                    _limit_units(units, members, "call", "%d haplotypes x ploidy %d: %s" % (H, K, e))
                    continue
                pending.append((model, handle, members))
        for model, handle, members in pending:
            for key, tr in zip(members, model.finish_batch_summaries(handle)):
                results[key] = tr
        for ri, unit in enumerate(units):
            rec, locus = unit["rec"], unit["locus"]
            H, M = locus.haplotypes.shape
            res = {}
            if unit["invalid"] is None:
                for s in samples:
                    K = int(ploidy_of(s))
                    if M == 0:
                        trace = GenotypeAllelesMultiTrace(np.zeros((chains, steps, K), np.int8), np.full((chains, steps), np.nan), 1)
                        summ = CallSummary.of_trace(trace.burn(burn), incongruence_threshold)
                    else:
                        summ = results[(ri, s)]
                    if not unit["keep"].all() and M > 0:
                        summ = summ.relabel(np.flatnonzero(unit["keep"]))
                    f0, _, o0 = summ.posterior_frequencies()  # over the alleles the trace knows: pad to the record's
                    freqs, occur = np.zeros(H), np.zeros(H)
                    freqs[: len(f0)], occur[: len(o0)] = f0[:H], o0[:H]
                    r = dict(alleles=np.asarray(summ.alleles), gprob=float(summ.gprob), sprob=float(summ.sprob), freqs=freqs, occur=occur,
                             mci=int(summ.mci))
                    if "GP" in report or "FORMAT/GP" in report:
                        r["GP"] = summ.posterior().as_array(H)
                    if "GL" in report or "FORMAT/GL" in report:
                        sr = unit["reads"][s]
                        r["GL"] = calling.genotype_likelihoods(sr["dists"], K, locus.haplotypes, read_counts=sr["counts"]).astype(np.float64) / np.log(10)
                    res[(ri, s)] = r
            flt, info, fmt, cols = _format_exact_record(dict(unit, invalid=unit["invalid"]), samples, _Forced(res), ri, ploidy_of, report, prior_frequencies_tag)
            # MCI is a sampler statistic here: per sample in the FORMAT column, the number of incongruent samples in INFO
            if unit["invalid"] is None:
                n_inc = 0
                for s in samples:
                    f = cols[s].split(":")
                    f[10] = str(res[(ri, s)]["mci"])
                    n_inc += int(res[(ri, s)]["mci"] > 0)
                    cols[s] = ":".join(f)
                info = info.replace(";MCI=0;", ";MCI=%d;" % n_inc)
            alts = unit["locus"].sequences[1:]  # (the input's, minus the alleles --filter-input-haplotypes removed)
            alt = ",".join(alts) if alts else "."
            yield "\t".join([rec["chrom"], str(rec["pos"]), rec["id"], rec["ref"], alt, ".", flt, info, fmt] + [cols[s] for s in samples])


_CALL_STREAMS = []


def _call_stream(i):
    """One of four side streams (made on first use) for the i-th batch of a block of records; None without a GPU runtime."""
    try:
        import torch

        if not torch.cuda.is_available():
            return None
        while len(_CALL_STREAMS) < 4:
            _CALL_STREAMS.append(torch.cuda.Stream())
        return _CALL_STREAMS[i % 4]
    except Exception:
        return None


class _Forced(dict):
    """results mapping for _format_exact_record that also serves records with a single haplotype or no position (the
    sampler path has its own summaries for those)."""
