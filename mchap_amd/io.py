"""Text / binary readers and locus objects for the application layer: a minimal BAM reader (no pysam), VCF and
BED4 text readers, the two locus types of the reference (io/loci.py: LocusPrior for `call-exact`, Locus + SNVs for
`assemble`), read extraction (io/bam.py:54-229) and the VCF value formatting of io/vcf/util.py / io/util.py.
Host code (numpy); the reference does the same work through pysam."""
import gzip
import struct
from itertools import combinations_with_replacement  # noqa: F401

import numpy as np

SEQ_CODE = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def read_sam(path):
    """-> (ref_names, {RG id: SM}, [record dicts]) of a SAM text file: the same records read_bam gives for the
    equivalent BAM (0-based pos, CIGAR as (length, op) pairs, phred qualities as ints, RG tag)."""
    refs, rg, recs = [], {}, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("@"):
                fields = line.split("\t")
                kv = dict(x.split(":", 1) for x in fields[1:] if ":" in x)
                if fields[0] == "@SQ":
                    refs.append(kv["SN"])
                elif fields[0] == "@RG":
                    rg[kv["ID"]] = kv["SM"]
                continue
            c = line.split("\t")
            cigar = []
            if c[5] != "*":
                n = ""
                for ch in c[5]:
                    if ch.isdigit():
                        n += ch
                    else:
                        cigar.append((int(n), ch))
                        n = ""
            seq = "" if c[9] == "*" else c[9]
            qual = [255] * len(seq) if c[10] == "*" else [ord(q) - 33 for q in c[10]]
            tag_rg = None
            for t in c[11:]:
                if t.startswith("RG:Z:"):
                    tag_rg = t[5:]
            recs.append(dict(qname=c[0], flag=int(c[1]), ref=None if c[2] == "*" else c[2], pos=int(c[3]) - 1, mapq=int(c[4]),
                             cigar=cigar, seq=seq, qual=qual, rg=tag_rg))
    return refs, rg, recs


def read_alignments(path, id_field="SM"):
    """read_bam or read_sam by what the file is (BAM files are gzip streams).  id_field: the read-group field that names
    the sample, "SM" or "ID" (--read-group-field; io/bam.py:21-51)."""
    assert id_field in ("SM", "ID")
    with open(path, "rb") as f:
        magic = f.read(2)
    refs, rg, recs = read_bam(path) if magic == b"\x1f\x8b" else read_sam(path)
    if id_field == "ID":
        rg = {k: k for k in rg}
    return refs, rg, recs


def read_bam(path):
    """-> (ref_names, {RG id: SM}, [record dicts]) of a (small) BAM file."""
    data = gzip.decompress(open(path, "rb").read())
    assert data[:4] == b"BAM\1"
    (l_text,) = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode().rstrip("\0")
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        o += 4
        refs.append(data[o:o + l_name - 1].decode())
        o += l_name + 4
    rg = {}
    for line in text.splitlines():
        if line.startswith("@RG"):
            f = dict(x.split(":", 1) for x in line.split("\t")[1:])
            rg[f["ID"]] = f["SM"]
    recs = []
    while o < len(data):
        (block,) = struct.unpack_from("<i", data, o)
        o += 4
        b = data[o:o + block]
        o += block
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        qname = b[p:p + l_name - 1].decode()
        p += l_name
        cigar = [(v >> 4, CIGAR_OPS[v & 15]) for v in struct.unpack_from("<%dI" % n_cig, b, p)]
        p += 4 * n_cig
        sq = b[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        seq = "".join(SEQ_CODE[(sq[i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = list(b[p:p + l_seq])
        p += l_seq
        tags = {}
        while p < len(b):
            tag, typ = b[p:p + 2].decode(), chr(b[p + 2])
            p += 3
            if typ == "Z":
                e = b.index(b"\0", p)
                tags[tag] = b[p:e].decode()
                p = e + 1
            elif typ == "A":
                tags[tag] = chr(b[p])
                p += 1
            elif typ in "cCsSiIf":
                n = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[typ]
                p += n
            elif typ == "B":
                sub = chr(b[p])
                (cnt,) = struct.unpack_from("<i", b, p + 1)
                p += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            else:
                raise ValueError(typ)
        recs.append(dict(qname=qname, flag=flag, ref=refs[ref_id] if ref_id >= 0 else None, pos=pos, mapq=mapq, cigar=cigar,
                         seq=seq, qual=qual, rg=tags.get("RG")))
    return refs, rg, recs


def open_text(path):
    """A text file that may be gzip / bgzip compressed (`.vcf.gz` inputs: a BGZF file is a multi-member gzip stream)."""
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path)


def vcf_contigs(path):
    """[(name, length)] of the ##contig lines of a VCF file (the header contigs of `call` / `call-exact`,
    application/baseclass.py:86-89)."""
    out = []
    with open_text(path) as f:
        for line in f:
            if not line.startswith("##"):
                break
            if line.startswith("##contig=<"):
                kv = dict(x.split("=", 1) for x in line.strip()[10:-1].split(",") if "=" in x)
                out.append((kv.get("ID"), int(kv.get("length", 0))))
    return out


def read_vcf(path):
    """-> (sample names, [record dicts]) of a VCF file (plain text or gzip / bgzip compressed)."""
    samples, out = [], []
    for line in open_text(path):
        line = line.rstrip("\n")
        if line.startswith("##") or not line:
            continue
        f = line.split("\t")
        if line.startswith("#"):
            samples = f[9:]
            continue
        info = {}
        for kv in f[7].split(";"):
            if "=" in kv:
                k, v = kv.split("=", 1)
                info[k] = v
            else:
                info[kv] = True
        out.append(dict(chrom=f[0], pos=int(f[1]), id=f[2], ref=f[3], alts=() if f[4] == "." else tuple(f[4].split(",")),
                        info=info, format=f[8] if len(f) > 8 else "", samples=dict(zip(samples, f[9:]))))
    return samples, out


_FILTER_OPS = {"=": np.equal, "==": np.equal, "!=": np.not_equal, ">": np.greater, ">=": np.greater_equal,
               "<": np.less, "<=": np.less_equal}


def parse_allele_filter(text):
    """`--filter-input-haplotypes` (application/arguments.py:389-401): '<INFO field><operator><number>' ->
    (field, comparison, number).  Operators = == != > >= < <=."""
    import re

    m = re.fullmatch(r"(\w+?)(==|!=|>=|<=|=|>|<)([0-9]*[.,]?[0-9]*)", text)
    if not m or m.group(3) in ("", ".", ","):
        raise ValueError("Invalid allele filter '%s'" % text)
    number = m.group(3).replace(",", ".")
    return m.group(1), _FILTER_OPS[m.group(2)], (int(number) if number.isdigit() else float(number))


def vcf_info_numbers(path):
    """{INFO field id: its Number} from the ##INFO lines of a VCF file."""
    out = {}
    for line in open_text(path):
        if not line.startswith("##"):
            break
        if line.startswith("##INFO=<"):
            kv = dict(x.split("=", 1) for x in line.strip()[8:-1].split(",") if "=" in x)
            out[kv.get("ID")] = kv.get("Number")
    return out


class Locus:
    """The locus of a record of known haplotypes (LocusPrior.from_variant_record, io/loci.py:193-303).

    allele_filter: (field, comparison, number) from parse_allele_filter plus the field's Number ('R': one value per
    allele, 'A': one per alternate allele) as a 4-tuple.  Alleles whose value fails the comparison are left out of the
    locus; a failing reference allele stays in the list but is masked (prior frequency 0) like a REFMASKED record; a
    record without the field keeps every allele (io/filter_alleles.py:56-95)."""

    def __init__(self, rec, frequency_tag=None, allele_filter=None):
        self.contig, self.start, self.stop = rec["chrom"], rec["pos"] - 1, rec["pos"] - 1 + len(rec["ref"])
        sequences = (rec["ref"],) + rec["alts"]
        n = len(sequences)
        self.mask_reference_allele = "REFMASKED" in rec["info"]
        keep = np.ones(n, dtype=bool)
        if allele_filter is not None:
            field, compare, number, length = allele_filter
            if length not in ("R", "A"):
                raise ValueError("Allele filter of field of invalid length '%s'" % length)
            if field in rec["info"] and rec["info"][field] is not True:
                values = np.array([float(x) if x != "." else np.nan for x in str(rec["info"][field]).split(",")])
                if length == "R":
                    assert len(values) == n
                    keep = np.asarray(compare(values, number), dtype=bool)
                else:
                    assert len(values) == n - 1
                    keep[1:] = compare(values, number)
            if not keep[0]:  # the reference allele is masked, not removed
                self.mask_reference_allele = True
                keep[0] = True
        if frequency_tag:
            fr = np.array([float(x) for x in rec["info"][frequency_tag].split(",")])
            assert len(fr) == n
        else:
            fr = np.ones(n) / n
        if self.mask_reference_allele:
            fr[0] = 0
        self.sequences = tuple(q for q, k in zip(sequences, keep) if k)
        fr = fr[keep]
        n = len(self.sequences)
        self.frequencies = fr / fr.sum() if fr.sum() > 0 else np.full(n, np.nan)
        haps = np.array([list(s) for s in self.sequences])
        offsets = np.where((haps != haps[0:1]).any(axis=0))[0]
        self.positions = [int(o) + self.start for o in offsets]
        self.alleles = []
        for o in offsets:
            col = haps[:, o]
            _, idx = np.unique(col, return_index=True)
            idx.sort()
            self.alleles.append(tuple(col[idx]))
        self.n_alleles = [len(a) for a in self.alleles]
        # encode_haplotypes (io/loci.py:184-191)
        self.haplotypes = np.zeros((n, len(offsets)), dtype=np.int8)
        for j, o in enumerate(offsets):
            lut = {c: i for i, c in enumerate(self.alleles[j])}
            self.haplotypes[:, j] = [lut[c] for c in haps[:, o]]


def extract_read_variants(locus, bam, sample, min_quality=20, skip_duplicates=True, skip_qcfail=True, skip_supplementary=True):
    """io/bam.py:54-229 for one sample: chars [n_reads, n_snv] ('-' gap, 'N' conflict) and summed quals.  `bam`: what
    read_alignments returned (its read-group table maps the group id to the sample field chosen there)."""
    _, rg, recs = bam
    skip = (0x400 if skip_duplicates else 0) | (0x200 if skip_qcfail else 0) | (0x800 if skip_supplementary else 0)
    positions = {p: i for i, p in enumerate(locus.positions)}
    n = len(locus.positions)
    data = {}
    for r in recs:
        if r["ref"] != locus.contig or r["flag"] & 4:
            continue
        ref_len = sum(l for l, op in r["cigar"] if op in "MDN=X")
        if not (r["pos"] < locus.stop and r["pos"] + ref_len > locus.start):
            continue  # pysam fetch(contig, start, stop): overlapping reads
        if r["mapq"] < min_quality or r["flag"] & skip:
            continue
        if rg.get(r["rg"]) != sample:
            continue
        if r["qname"] not in data:
            data[r["qname"]] = [np.full(n, "-", dtype="U1"), np.zeros(n, dtype=np.int16)]
        chars, quals = data[r["qname"]]
        rp, gp = 0, r["pos"]
        for l, op in r["cigar"]:
            if op in "M=X":
                for k in range(l):
                    if gp + k in positions:
                        i = positions[gp + k]
                        c, q = r["seq"][rp + k], r["qual"][rp + k]
                        if chars[i] == "-":
                            chars[i], quals[i] = c, q
                        elif chars[i] == c:
                            quals[i] += q
                        else:
                            chars[i] = "N"
                rp += l
                gp += l
            elif op in "IS":
                rp += l
            elif op in "DN":
                gp += l
    if not data:
        return np.empty((0, n), dtype="U1"), np.empty((0, n), dtype=np.int16)
    return np.array([v[0] for v in data.values()]), np.array([v[1] for v in data.values()])


def qual_of_prob(prob, precision=6):
    """Phred-scaled quality of a probability of being right, as the VCF fields GQ / SQ want it (io/util.py:56-89): the
    probability is cut off (not rounded) after `precision` decimals and capped one unit below 1, so the quality is finite."""
    unit = 10 ** precision
    kept = np.floor(min(prob, 1 - 0.1 ** precision) * unit) / unit
    return int(np.round(-10 * np.log10(1 - kept)))


def _number_text(x, precision, negative_zero):
    """One float of a VCF field: rounded, without a trailing '.0', '.' for NaN."""
    if np.isnan(x):
        return "."
    r = float(np.round(x, precision))
    if np.isfinite(r) and r == int(r):
        return "-0" if (negative_zero and r == 0 and np.signbit(r)) else str(int(r))
    return repr(r)[:16]


def vcfstr(obj, precision=3):
    """The text of a VCF value (io/vcf/util.py:4-42): None / NaN / an empty array -> '.', arrays comma-separated, floats rounded
    to `precision` decimals and whole numbers written without a fraction."""
    if obj is None:
        return "."
    if isinstance(obj, np.ndarray):
        if obj.size == 0:
            return "."
        if np.issubdtype(obj.dtype, np.floating):
            return ",".join(_number_text(x, precision, True) for x in obj.tolist())
        return ",".join(str(x) for x in obj.tolist())
    if isinstance(obj, (float, np.floating)):
        return _number_text(float(obj), precision, False)
    return str(obj)


# ---- de novo assembly loci (mchap assemble): targets from a BED file, SNVs from a VCF (io/loci.py:94-135) ----
class DenovoLocus:
    def __init__(self, contig, start, stop, name, variant_records, sequence, sequence_known=True):
        """sequence_known=False: `sequence` came from a reference known by its index only (io.Reference without the FASTA: all
        'N'); the variants' REF alleles are then written at their positions.  A real FASTA's own N bases are never touched."""
        self.contig, self.start, self.stop, self.name, self.sequence = contig, start, stop, name, sequence
        snps = {}
        for r in variant_records:
            alleles = (r["ref"],) + r["alts"]
            if r["chrom"] != contig or not (start <= r["pos"] - 1 < stop) or any(len(a) != 1 for a in alleles):
                continue
            p = r["pos"] - 1
            if p in snps:  # _merge_snps (io/loci.py:364-382)
                assert snps[p][0] == alleles[0]
                snps[p] = snps[p] + tuple(a for a in alleles if a not in snps[p])
            else:
                snps[p] = alleles
        self.positions = list(snps)
        self.alleles = [snps[p] for p in self.positions]
        self.n_alleles = [len(a) for a in self.alleles]
        if self.sequence is not None and not sequence_known:
            chars = list(self.sequence)
            for p, tup in zip(self.positions, self.alleles):
                if chars[p - start] == "N":
                    chars[p - start] = tup[0]
            self.sequence = "".join(chars)

    def format_haplotype(self, alleles):
        chars = list(self.sequence)
        for p, tup, a in zip(self.positions, self.alleles, alleles):
            chars[p - self.start] = tup[int(a)]
        return "".join(chars)


def read_bed4(path):
    return [(f[0], int(f[1]), int(f[2]), f[3]) for f in (line.split() for line in open(path)) if len(f) >= 4]


# ---- inputs of the command line programs ----
def read_fasta(path):
    """{contig: sequence} of a (possibly gzip / bgzip compressed) FASTA file; the fetch of io/loci.py:86-92 is a slice of it."""
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    seqs, name, parts = {}, None, []
    with opener(path, "rt") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.upper())
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


def _is_sam(path):
    """A SAM text file: not gzip, and its first line is a header line (or an alignment line of >= 11 fields)."""
    try:
        with open(path, "rb") as f:
            head = f.read(4096)
    except OSError:
        return False
    if head[:2] == b"\x1f\x8b" or not head:
        return False
    try:
        first = head.decode("ascii").splitlines()[0]
    except (UnicodeDecodeError, IndexError):
        return False
    return first.startswith(("@HD", "@SQ", "@RG", "@PG", "@CO")) or len(first.split("\t")) >= 11


def bam_header(path):
    """(reference names with lengths, {read group id: sample}) of a BAM file (or of a SAM text file)."""
    if _is_sam(path):
        refs, rg = [], {}
        for line in open(path):
            if not line.startswith("@"):
                break
            f = line.rstrip("\n").split("\t")
            kv = dict(x.split(":", 1) for x in f[1:] if ":" in x)
            if f[0] == "@SQ":
                refs.append((kv["SN"], int(kv.get("LN", 0))))
            elif f[0] == "@RG":
                rg[kv["ID"]] = kv.get("SM", kv["ID"])
        return refs, rg
    data = gzip.open(path, "rb")
    head = data.read(8)
    assert head[:4] == b"BAM\1"
    (l_text,) = struct.unpack("<i", head[4:8])
    text = data.read(l_text).decode().rstrip("\0")
    (n_ref,) = struct.unpack("<i", data.read(4))
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack("<i", data.read(4))
        nm = data.read(l_name)[:-1].decode()
        (l_ref,) = struct.unpack("<i", data.read(4))
        refs.append((nm, l_ref))
    rg = {}
    for line in text.splitlines():
        if line.startswith("@RG"):
            f = dict(x.split(":", 1) for x in line.split("\t")[1:])
            rg[f["ID"]] = f.get("SM", f["ID"])
    return refs, rg


def sample_bam_table(bam_args, id_field="SM"):
    """--bam: (1) BAM paths, (2) a text file with one path per line, (3) a text file of `sample<TAB>path` lines
    (reference application/arguments.py:135-152, 890-955).  Returns an ordered {sample: path}: for (1) and (2) every
    sample of every BAM's read groups (named by their `id_field`, --read-group-field)."""
    paths, table = [], {}

    def groups(p):
        rg = bam_header(p)[1]
        return list(rg) if id_field == "ID" else list(rg.values())

    if len(bam_args) == 1 and not _is_bam(bam_args[0]) and not _is_sam(bam_args[0]):
        for line in open(bam_args[0]):
            f = line.rstrip("\n").split("\t")
            if not f or not f[0].strip():
                continue
            if len(f) == 1:
                paths.append(f[0].strip())
            else:
                if f[0] in table:
                    raise IOError('Duplicate input sample name "%s"' % f[0])
                table[f[0]] = f[1].strip()
        if table:
            for s, p in table.items():
                if s not in groups(p):
                    raise IOError('Sample "%s" was not found in bam "%s"' % (s, p))
            return table
    else:
        paths = list(bam_args)
    for p in paths:
        for s in dict.fromkeys(groups(p)):
            if s in table:
                raise IOError('Duplicate input sample name "%s"' % s)
            table[s] = p
    return table


def sample_pools(sample_bams, pool_arg):
    """--sample-pool (application/arguments.py:848-887): None -> every sample is its own pool; a name -> one pool of all
    samples; a file of `sample<TAB>pool` lines -> custom pools (a sample may be in several).  Returns an ordered
    {pool: [(sample, bam path), ...]}: the pool names take the place of the sample names everywhere else."""
    import os

    if pool_arg is None:
        return {s: [(s, p)] for s, p in sample_bams.items()}
    if not os.path.isfile(pool_arg):
        return {pool_arg: list(sample_bams.items())}
    pools, seen = {}, set()
    for line in open(pool_arg):
        f = line.strip().split("\t")
        if len(f) < 2:
            continue
        sample, pool = f[0], f[1]
        seen.add(sample)
        if sample not in sample_bams:
            raise ValueError("The following names in the sample-pool file do not match a known sample : {%r}" % sample)
        pools.setdefault(pool, []).append((sample, sample_bams[sample]))
    missing = set(sample_bams) - seen
    if missing:
        raise ValueError("The following samples have not been assigned to a pool: %s" % sorted(missing))
    return pools


def sample_temperatures(args, samples):
    """--mcmc-temperatures (application/arguments.py:1122-1166): a list of inverse temperatures for every sample, or a file
    of `sample<TAB>t1<TAB>t2...` lines (samples not listed: no tempering).  Sorted, 1.0 appended when missing.
    Returns {sample: tuple of floats}."""
    def ladder(values):
        temps = sorted(float(v) for v in values)
        assert temps[0] > 0.0 and temps[-1] <= 1.0
        if temps[-1] != 1.0:
            temps.append(1.0)
        return tuple(temps)

    args = [str(a) for a in args]
    if len(args) > 1 or args[0].replace(".", "", 1).isdigit():
        t = ladder(args)
        return {s: t for s in samples}
    data = {s: (1.0,) for s in samples}
    for line in open(args[0]):
        f = line.strip().split("\t")
        if len(f) >= 2:
            data[f[0]] = ladder(f[1:])
    assert len(data) == len(samples), "a sample of the temperatures file is not an input sample"
    return data


def parse_region(text):
    """--region contig:start-stop (io/loci.py:161-172: the numbers are used as a 0-based half-open interval, like a BED line)."""
    contig, interval = text.strip().split(":")
    start, stop = interval.strip().split("-")
    return contig, int(start), int(stop)


class Reference:
    """The reference genome of `mchap assemble`: contig lengths and sequence slices (io/loci.py:86-92).  From a FASTA file
    (plain or gzip / bgzip compressed); or, when only its `.fai` index is at hand, the contig lengths with every base
    unknown (`N`): the assembled haplotypes then carry the variants' own alleles at the SNV positions and N elsewhere."""

    def __init__(self, path, allow_index_only=False):
        """allow_index_only: accept a FASTA of which only the `.fai` index exists (contig lengths for the header, 'N' for every
        base that is not an SNV).  Off by default: a mistyped path must not silently produce records with N sequences."""
        import os

        self.path = path
        self.known = os.path.isfile(path)
        if self.known:
            self.seqs = read_fasta(path)
            self.contigs = [(n, len(q)) for n, q in self.seqs.items()]
        else:
            fai = path + ".fai"
            if not os.path.isfile(fai):
                raise IOError("reference '%s' not found (nor its index '%s')" % (path, fai))
            if not allow_index_only:
                raise IOError("reference '%s' not found; only its index '%s' exists (pass --reference-index-only to write N for the "
                              "unknown reference bases)" % (path, fai))
            self.seqs = None
            self.contigs = [(f[0], int(f[1])) for f in (line.split("\t") for line in open(fai)) if len(f) >= 2]
            self.lengths = dict(self.contigs)

    def fetch(self, contig, start, stop):
        if self.known:
            return self.seqs[contig][start:stop]
        n = self.lengths[contig]
        return "N" * (max(0, min(stop, n) - max(0, start)))


def _is_bam(path):
    try:
        with gzip.open(path, "rb") as f:
            return f.read(4) == b"BAM\1"
    except OSError:
        return False


def sample_values(arg, samples, cast, default=None):
    """--ploidy / inbreeding: one value for all samples, or a text file of `sample<TAB>value` lines naming every sample
    (reference application/arguments.py:957-988, 1122-1166).  Returns a value or a {sample: value} mapping."""
    import os

    if arg is None:
        return default
    if os.path.isfile(str(arg)):
        table = {}
        for line in open(arg):
            f = line.rstrip("\n").split("\t")
            if len(f) >= 2:
                table[f[0]] = cast(f[1])
        missing = [s for s in samples if s not in table]
        if missing:
            raise IOError('Sample "%s" is not specified in "%s"' % (missing[0], arg))
        return {s: table[s] for s in samples}
    return cast(arg)


# ---------------------------------------------------------------------------------------------------------------
# Alignment files at scale (reference io/bam.py:54-229 fetches by region through pysam): BGZF blocks inflated one by
# one (zlib releases the interpreter lock: a thread pool inflates in parallel), the `.bai` index for region fetches, records
# kept as columns (numpy), and extract_read_variants vectorised over a locus's records instead of one Python loop per read.
# ---------------------------------------------------------------------------------------------------------------
def bgzf_blocks(data, first=0, count=None, last=None):
    """[(offset, compressed size, header size)] of the BGZF blocks of a byte string or mapped file (SAM/BAM specification 4.1: a
    gzip member whose extra field carries its own size: BSIZE), from the block at file offset `first`; at most `count` blocks, and
    none that starts beyond file offset `last`."""
    out, o, n = [], first, len(data)
    while o + 18 <= n and (count is None or len(out) < count) and (last is None or o <= last):
        if data[o:o + 4] != b"\x1f\x8b\x08\x04":
            raise IOError("not a BGZF block at offset %d" % o)
        (xlen,) = struct.unpack_from("<H", data, o + 10)
        p, end, bsize = o + 12, o + 12 + xlen, None
        while p + 4 <= end:
            si1, si2, slen = data[p], data[p + 1], struct.unpack_from("<H", data, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, p + 4)[0] + 1
            p += 4 + slen
        if bsize is None:
            raise IOError("gzip member without a BGZF size field at offset %d" % o)
        out.append((o, bsize, 12 + xlen))
        o += bsize
    return out


def bgzf_inflate(data, blocks, workers=1):
    """The inflated payloads of the given blocks, in order."""
    import zlib

    def one(b):
        o, size, head = b
        return zlib.decompress(data[o + head:o + size - 8], -15)

    if workers > 1 and len(blocks) > 8:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(one, blocks))
    return [one(b) for b in blocks]


def reg2bins(beg, end):
    """The bins of the UCSC binning scheme that may hold records overlapping [beg, end) (SAM specification 5.3)."""
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


def read_bai(path):
    """-> per reference: ({bin: [(chunk begin, chunk end) virtual offsets]}, linear index array)."""
    data = open(path, "rb").read()
    assert data[:4] == b"BAI\1"
    (n_ref,) = struct.unpack_from("<i", data, 4)
    o, refs = 8, []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", data, o)
        o += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", data, o)
            o += 8
            chunks = np.frombuffer(data, dtype="<u8", count=2 * n_chunk, offset=o).reshape(n_chunk, 2)
            o += 16 * n_chunk
            bins[b] = chunks
        (n_intv,) = struct.unpack_from("<i", data, o)
        o += 4
        lin = np.frombuffer(data, dtype="<u8", count=n_intv, offset=o)
        o += 8 * n_intv
        refs.append((bins, lin))
    return refs


_AUX_SIZE = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}


def _aux_rg(buf, o, end):
    """The value of a record's RG:Z tag found by walking its aux fields (SAM specification 4.2.4), or None -- on malformed aux data
    exactly what the native walk gives (csrc/bam_columns.cpp find_rg): None for an unterminated string, an unknown type or B
    sub-type, a negative count, a truncated B header -- so that a sample's reads do not depend on which of the two walks ran."""
    while o + 3 <= end:
        tag, typ = bytes(buf[o:o + 2]), chr(buf[o + 2])
        o += 3
        if typ in ("Z", "H"):
            z = o
            while z < end and buf[z] != 0:
                z += 1
            if z >= end:
                return None  # no terminating NUL
            if tag == b"RG" and typ == "Z":
                return bytes(buf[o:z]).decode()
            o = z + 1
        elif typ == "B":
            if o + 5 > end:
                return None
            size = _AUX_SIZE.get(chr(buf[o]))
            (cnt,) = struct.unpack_from("<i", buf, o + 1)
            if size is None or cnt < 0:
                return None
            o += 5 + cnt * size
        elif typ in _AUX_SIZE:
            o += _AUX_SIZE[typ]
        else:
            return None
    return None


def _aux_rg_all(b, tag_off, rec_end):
    """_aux_rg for every record at once: the records walk their aux fields in lockstep (one numpy step per field index).  Returns
    (start, length) of each record's RG:Z value in `b`, start -1 where the walk finds none or gives up."""
    n = len(tag_off)
    start, length = np.full(n, -1, dtype=np.int64), np.zeros(n, dtype=np.int64)
    if n == 0:
        return start, length
    size_of = np.zeros(256, dtype=np.int64)
    for k, v in _AUX_SIZE.items():
        size_of[ord(k)] = v
    nul = np.flatnonzero(b == 0)
    p = np.asarray(tag_off, dtype=np.int64).copy()
    end = np.asarray(rec_end, dtype=np.int64)
    act = np.flatnonzero(p + 3 <= end)
    while len(act):
        q, e = p[act], end[act]
        t0, t1, ty = b[q], b[q + 1], b[q + 2]
        q = q + 3
        nxt = np.full(len(act), -1, dtype=np.int64)  # -1: the walk of this record is over
        is_s = (ty == ord("Z")) | (ty == ord("H"))
        if is_s.any():
            qs, es = q[is_s], e[is_s]
            j = np.searchsorted(nul, qs)  # the first NUL at or after the value's start
            z = np.where(j < len(nul), nul[np.minimum(j, max(len(nul) - 1, 0))] if len(nul) else 0, np.int64(1) << 62)
            closed = z < es
            hit = closed & (t0[is_s] == ord("R")) & (t1[is_s] == ord("G")) & (ty[is_s] == ord("Z"))
            rows = act[is_s]
            start[rows[hit]], length[rows[hit]] = qs[hit], (z - qs)[hit]
            nxt[is_s] = np.where(closed & ~hit, z + 1, -1)
        is_b = ty == ord("B")
        if is_b.any():
            qb = q[is_b]
            room = qb + 5 <= e[is_b]
            at = np.where(room, qb, 0)
            at = np.minimum(at, max(len(b) - 5, 0))
            size = size_of[b[at]]
            cnt = np.ascontiguousarray(b[(at + 1)[:, None] + np.arange(4)]).view("<i4").reshape(-1).astype(np.int64) if len(b) >= 5 else np.zeros(len(at), np.int64)
            nxt[is_b] = np.where(room & (size > 0) & (cnt >= 0), qb + 5 + cnt * size, -1)
        is_f = ~is_s & ~is_b
        if is_f.any():
            size = size_of[ty[is_f]]
            nxt[is_f] = np.where(size > 0, q[is_f] + size, -1)
        go = (nxt >= 0) & (nxt + 3 <= e)
        p[act[go]] = nxt[go]
        act = act[go]
    return start, length


class AlignmentColumns:
    """The records of (a region of) a BAM file as columns: ref_id, pos, end, mapq, flag, read-group index, query-name id,
    and the flattened CIGAR operations / packed sequences / qualities they index into."""

    def __init__(self, refs, rg_table, buf, offsets, id_field="SM"):
        self.refs = refs
        n = len(offsets)
        self.n = n
        offs = np.asarray(offsets, dtype=np.int64)
        b = np.frombuffer(buf, dtype=np.uint8)
        core = b[offs[:, None] + np.arange(36)] if n else np.zeros((0, 36), np.uint8)

        def field(lo, dt):
            return np.ascontiguousarray(core[:, lo:lo + np.dtype(dt).itemsize]).view(dt).reshape(n)

        block = field(0, "<i4")
        self.ref_id, self.pos = field(4, "<i4"), field(8, "<i4")
        l_name, self.mapq = core[:, 12].astype(np.int64), core[:, 13].astype(np.int32)
        n_cig, self.flag, l_seq = field(16, "<u2").astype(np.int64), field(18, "<u2").astype(np.int32), field(20, "<i4").astype(np.int64)
        name_off = offs + 36
        cig_off = name_off + l_name
        self.seq_off = cig_off + 4 * n_cig
        self.qual_off = self.seq_off + (l_seq + 1) // 2
        tag_off = self.qual_off + l_seq
        rec_end = offs + 4 + block
        self.buf = b
        # query names -> integer ids (mates share a name): the names as one fixed-width byte-string column
        if n:
            width = int(l_name.max())
            k_ = np.arange(width)
            nm = np.where(k_[None, :] < (l_name - 1)[:, None], b[np.minimum(name_off[:, None] + k_[None, :], len(b) - 1)], 0).astype(np.uint8)
            self.qname = np.unique(np.ascontiguousarray(nm).view("S%d" % width).reshape(n), return_inverse=True)[1].astype(np.int64)
        else:
            self.qname = np.zeros(0, dtype=np.int64)
        # read groups: the RG:Z field found by walking each record's aux fields in order (what pysam's read.get_tag("RG") does;
        # "RGZ" bytes inside another field's payload are not a field), its value matched with the header's read-group ids
        rg_names = list(rg_table)
        self.rg_samples = [rg_table[k] if id_field == "SM" else k for k in rg_names]
        rgi = np.full(n, -1, dtype=np.int64)
        if n and rg_names:
            v0, vlen = _aux_rg_all(b, tag_off, rec_end)
            for gi, key in enumerate(rg_names):
                kb = np.frombuffer(key.encode(), dtype=np.uint8)
                m = (v0 >= 0) & (vlen == len(kb)) & (rgi < 0)
                if len(kb):
                    m[m] = (b[v0[m][:, None] + np.arange(len(kb))] == kb).all(axis=1)
                rgi[m] = gi
        self.rg = rgi
        # CIGAR operations, flattened: record, op, length, reference / read offset at the start of the op
        total = int(n_cig.sum())
        rec = np.repeat(np.arange(n), n_cig)
        first = np.repeat(np.cumsum(n_cig) - n_cig, n_cig)
        k = np.arange(total) - first
        words = b[(np.repeat(cig_off, n_cig) + 4 * k)[:, None] + np.arange(4)] if total else np.zeros((0, 4), np.uint8)
        w = np.ascontiguousarray(words).view("<u4").reshape(total)
        op, ln = (w & 15).astype(np.int64), (w >> 4).astype(np.int64)
        ref_adv = np.where(np.isin(op, (0, 2, 3, 7, 8)), ln, 0)   # M D N = X consume the reference
        read_adv = np.where(np.isin(op, (0, 1, 4, 7, 8)), ln, 0)  # M I S = X consume the read
        cref, cread = np.cumsum(ref_adv) - ref_adv, np.cumsum(read_adv) - read_adv
        base_ref = np.repeat((np.cumsum(np.bincount(rec, ref_adv, n)) - np.bincount(rec, ref_adv, n)) if total else np.zeros(n), n_cig)
        base_read = np.repeat((np.cumsum(np.bincount(rec, read_adv, n)) - np.bincount(rec, read_adv, n)) if total else np.zeros(n), n_cig)
        self.c_rec, self.c_op, self.c_len = rec, op, ln
        self.c_ref0 = self.pos[rec].astype(np.int64) + (cref - base_ref).astype(np.int64)
        self.c_read0 = (cread - base_read).astype(np.int64)
        self.end = self.pos.astype(np.int64) + np.bincount(rec, ref_adv, n).astype(np.int64) if total else self.pos.astype(np.int64)
        # coordinate-sorted files (the usual case, and what an index requires): the records that can overlap a region are
        # a contiguous run found by bisection; the CIGAR operations of a run of records are a contiguous run as well
        self.seg_first = np.r_[np.cumsum(n_cig) - n_cig, total].astype(np.int64)
        rid = np.where(self.ref_id < 0, np.int64(1) << 30, self.ref_id.astype(np.int64))  # unplaced reads sort last
        self.sort_key = (rid << 32) | self.pos.astype(np.int64).clip(0)
        self.sorted = bool(n < 2 or (self.sort_key[1:] >= self.sort_key[:-1]).all())
        self.max_span = int((self.end - self.pos).max()) if n else 0

    @classmethod
    def native(cls, refs, rg_table, buf, start, id_field="SM"):
        """The same table from the library's record walk (mchap_bam_columns, include/mchap_hip.h: one pass over the bytes in
        C++) for the records of `buf` from byte `start`; None when the library cannot be loaded (the constructor's array
        operations are the definition: tests/test_io_bam_matrices.py holds the two against each other)."""
        try:
            from . import _lib

            L = _lib.lib()
        except Exception:  # noqa: BLE001 -- no library: the numpy construction
            return None
        import ctypes as C

        b = np.frombuffer(buf, dtype=np.uint8)
        base = b.ctypes.data if len(b) else 0
        ops = C.c_int64(0)
        n = int(L.mchap_bam_count(C.c_void_p(base), len(b), int(start), C.byref(ops)))
        total = int(ops.value)
        self = cls.__new__(cls)
        self.refs, self.n, self.buf = refs, n, b
        i8 = lambda k: np.empty(k, dtype=np.int64)  # noqa: E731
        i4 = lambda k: np.empty(k, dtype=np.int32)  # noqa: E731
        offs, self.ref_id, self.pos, self.end, self.mapq, self.flag = i8(n), i4(n), i4(n), i8(n), i4(n), i4(n)
        self.seq_off, self.qual_off, self.rg, self.qname, self.seg_first = i8(n), i8(n), i8(n), i8(n), i8(n + 1)
        self.c_rec, self.c_op, self.c_len, self.c_ref0, self.c_read0 = i8(total), i8(total), i8(total), i8(total), i8(total)
        rg_names = list(rg_table)
        self.rg_samples = [rg_table[k] if id_field == "SM" else k for k in rg_names]
        ids = b"".join(k.encode() + b"\0" for k in rg_names) + b"\0"
        rc = L.mchap_bam_columns(C.c_void_p(base), len(b), int(start), n, ids, len(rg_names), *(C.c_void_p(a.ctypes.data) for a in (
            offs, self.ref_id, self.pos, self.end, self.mapq, self.flag, self.seq_off, self.qual_off, self.rg, self.qname, self.seg_first,
            self.c_rec, self.c_op, self.c_len, self.c_ref0, self.c_read0)))
        if rc != 0:
            raise ValueError("malformed BAM record")
        rid = np.where(self.ref_id < 0, np.int64(1) << 30, self.ref_id.astype(np.int64))  # unplaced reads sort last
        self.sort_key = (rid << 32) | self.pos.astype(np.int64).clip(0)
        self.sorted = bool(n < 2 or (self.sort_key[1:] >= self.sort_key[:-1]).all())
        self.max_span = int((self.end - self.pos).max()) if n else 0
        return self

    def window(self, tid, start, stop):
        """[lo, hi): the run of records that can overlap [start, stop) of reference `tid` (all records of an unsorted file)."""
        if not self.sorted or tid < 0:
            return 0, self.n
        base = np.int64(tid) << 32
        lo = int(np.searchsorted(self.sort_key, base | max(0, start - self.max_span), side="left"))
        hi = int(np.searchsorted(self.sort_key, base | max(0, stop), side="left"))
        return lo, hi


class BamFile:
    """A BAM file read block by block: the whole file (`columns()`), or -- when its `.bai` index is present -- only the BGZF
    blocks that can hold records of a region (`columns(contig, start, stop)`), as AlignmentColumns."""

    def __init__(self, path, id_field="SM", workers=1):
        import os

        import mmap

        self.path, self.id_field, self.workers = path, id_field, workers
        # the file is mapped, not read: a region fetch touches only the pages of the blocks it inflates, so N samples of
        # multi-GB files cost N x (a few blocks) of resident memory, as pysam's fetch does
        with open(path, "rb") as f:
            self.data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if os.fstat(f.fileno()).st_size else b""
        self._blocks = None
        first = bgzf_blocks(self.data, 0, 1)
        head = b"".join(bgzf_inflate(self.data, first))
        nxt = first[0][0] + first[0][1]
        while True:  # the header may span blocks
            try:
                self._parse_header(head)
                break
            except (struct.error, IndexError):
                blk = bgzf_blocks(self.data, nxt, 1)
                head += bgzf_inflate(self.data, blk)[0]
                nxt = blk[0][0] + blk[0][1]
        self.index = read_bai(path + ".bai") if os.path.isfile(path + ".bai") else None
        self._all = None
        self._last = (None, None)  # the columns of the last region fetched: all samples of a locus share one inflate + parse

    @property
    def blocks(self):
        """Every block of the file (walks all block headers: only the whole-file path needs it)."""
        if self._blocks is None:
            self._blocks = bgzf_blocks(self.data)
        return self._blocks

    def _parse_header(self, d):
        assert d[:4] == b"BAM\1"
        (l_text,) = struct.unpack_from("<i", d, 4)
        text = d[8:8 + l_text].decode().rstrip("\0")
        o = 8 + l_text
        (n_ref,) = struct.unpack_from("<i", d, o)
        o += 4
        refs = []
        for _ in range(n_ref):
            (l_name,) = struct.unpack_from("<i", d, o)
            nm = d[o + 4:o + 4 + l_name - 1].decode()
            (l_ref,) = struct.unpack_from("<i", d, o + 4 + l_name)
            refs.append((nm, l_ref))
            o += 8 + l_name
        self.refs, self.header_end = refs, o
        self.rg = {}
        for line in text.splitlines():
            if line.startswith("@RG"):
                f = dict(x.split(":", 1) for x in line.split("\t")[1:])
                self.rg[f["ID"]] = f.get("SM", f["ID"])

    def _columns_of(self, payload, start):
        cols = AlignmentColumns.native(self.refs, self.rg, payload, start, self.id_field)
        if cols is not None:
            return cols
        offsets, o, n = [], start, len(payload)
        while o + 4 <= n:
            (block,) = struct.unpack_from("<i", payload, o)
            if o + 4 + block > n:
                break
            offsets.append(o)
            o += 4 + block
        return AlignmentColumns(self.refs, self.rg, payload, offsets, self.id_field)

    def columns(self, contig=None, start=None, stop=None):
        if contig is None or self.index is None:
            if self._all is None:
                payload = b"".join(bgzf_inflate(self.data, self.blocks, self.workers))
                self._all = self._columns_of(payload, self.header_end)
            return self._all
        tid = [n for n, _ in self.refs].index(contig)
        bins, lin = self.index[tid]
        min_off = int(lin[min(start >> 14, len(lin) - 1)]) if len(lin) else 0
        chunks = [c for b in reg2bins(start, stop) if b in bins for c in bins[b] if int(c[1]) > min_off]
        if not chunks:
            return AlignmentColumns(self.refs, self.rg, b"", [], self.id_field)
        lo = min(int(c[0]) for c in chunks)
        hi = max(int(c[1]) for c in chunks)
        key = (lo, hi)
        if self._last[0] != key:
            # (virtual offsets are block file offsets << 16: the region's blocks are walked from the first one, nothing else is touched)
            payload = b"".join(bgzf_inflate(self.data, bgzf_blocks(self.data, lo >> 16, last=hi >> 16), self.workers))
            self._last = (key, self._columns_of(payload, lo & 0xFFFF))
        return self._last[1]


_NIB = np.frombuffer(SEQ_CODE.encode(), dtype=np.uint8)


def extract_read_variants_columns(locus, cols, sample, min_quality=20, skip_duplicates=True, skip_qcfail=True, skip_supplementary=True,
                                  as_codes=False):
    """extract_read_variants (io/bam.py:54-229) over AlignmentColumns, vectorised: same matrices as the per-read loop (rows in
    order of the first passing record of each query name; a position covered by both mates keeps the base when they agree,
    with the qualities added, and becomes 'N' when they do not).  Only the records of the locus's window are looked at
    (AlignmentColumns.window).  as_codes: the characters as uint8 ASCII codes instead of a 'U1' array."""
    n_snv = len(locus.positions)

    def result(chars, quals):
        return (chars if as_codes else chars.view("S1").astype("U1").reshape(chars.shape)), quals.astype(np.int16)

    skip = 0x4 | (0x400 if skip_duplicates else 0) | (0x200 if skip_qcfail else 0) | (0x800 if skip_supplementary else 0)
    names = [n for n, _ in cols.refs]
    tid = names.index(locus.contig) if cols.n and locus.contig in names else -1
    lo, hi = cols.window(tid, locus.start, locus.stop) if cols.n else (0, 0)
    w = slice(lo, hi)
    want_rg = np.array([s == sample for s in cols.rg_samples] + [False])  # (index -1: no read group)
    ok = (cols.ref_id[w] == tid) & (cols.pos[w] < locus.stop) & (cols.end[w] > locus.start) & ((cols.flag[w] & skip) == 0) & \
        (cols.mapq[w] >= min_quality) & want_rg[cols.rg[w]]
    recs = np.flatnonzero(ok)  # (relative to lo)
    if len(recs) == 0:
        return result(np.empty((0, n_snv), dtype=np.uint8), np.empty((0, n_snv), dtype=np.int16))
    # rows in order of the first passing record of each query name
    _, first_of, inv = np.unique(cols.qname[w][recs], return_index=True, return_inverse=True)
    order = np.argsort(first_of, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    rec_row = np.full(hi - lo, -1, dtype=np.int64)
    rec_row[recs] = rank[inv]
    chars = np.full((len(order), n_snv), ord("-"), dtype=np.uint8)
    quals = np.zeros((len(order), n_snv), dtype=np.int64)
    if n_snv:
        s0, s1 = int(cols.seg_first[lo]), int(cols.seg_first[hi])
        c_rec, c_op = cols.c_rec[s0:s1], cols.c_op[s0:s1]
        seg = np.flatnonzero(ok[c_rec - lo] & ((c_op == 0) | (c_op == 7) | (c_op == 8))) + s0
        P = np.asarray(locus.positions, dtype=np.int64)
        r0 = cols.c_ref0[seg]
        hit = (r0[:, None] <= P[None, :]) & (P[None, :] < (r0 + cols.c_len[seg])[:, None])
        si, pj = np.nonzero(hit)                   # (segment, SNV) pairs in record order, then position order
        s_ = seg[si]
        rec = cols.c_rec[s_]
        ro = cols.c_read0[s_] + (P[pj] - cols.c_ref0[s_])
        byte = cols.buf[cols.seq_off[rec] + (ro >> 1)]
        base = _NIB[np.where(ro & 1, byte & 15, byte >> 4)]
        q = cols.buf[cols.qual_off[rec] + ro].astype(np.int64)
        rows = rec_row[rec - lo]
        # a (row, SNV) cell hit more than once (overlapping mates): apply the hits in record order
        cell = rows * n_snv + pj
        o = np.argsort(cell, kind="stable")
        sc = cell[o]
        if len(sc) and (sc[1:] != sc[:-1]).all():  # every cell hit once (no overlapping mates): one scatter
            chars[rows, pj] = base
            quals[rows, pj] = q
        else:
            rank_c = np.zeros(len(cell), dtype=np.int64)
            startg = np.r_[True, sc[1:] != sc[:-1]]
            rank_c[o] = np.arange(len(sc)) - np.maximum.accumulate(np.where(startg, np.arange(len(sc)), 0))
            for k in range(int(rank_c.max(initial=-1)) + 1):
                m = rank_c == k
                r_, j_, b_, q_ = rows[m], pj[m], base[m], q[m]
                cur = chars[r_, j_]
                empty = cur == ord("-")
                same = cur == b_
                chars[r_, j_] = np.where(empty, b_, np.where(same, cur, ord("N")))
                quals[r_, j_] = np.where(empty, q_, np.where(same, quals[r_, j_] + q_, quals[r_, j_]))
    return result(chars, quals)
