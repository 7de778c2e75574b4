"""Synthetic loci for benchmarks and parity tests.

Restates what the reference's test-only simulator does (mchap/testing.py:9-73: sample read
haplotypes, assign qualities, encode with as_probabilistic, resample errors) with a per-unit
deterministic generator so that every rank / shard can build its own units.
"""
import numpy as np

PFEIFFER_ERROR = 0.0024  # reference mchap/constant.py:3


def as_probabilistic(calls, n_alleles, p, error_factor=3.0):
    """int8 calls [R, M] (+ per-call probability p) -> float64 tensor [R, M, A]
    (reference encoding/integer/transcode.py:16-77): called allele p, others (1-p)/3, alleles >= n_alleles[j]
    zero, gaps (call < 0) NaN *before* the zero mask."""
    calls = np.asarray(calls)
    n_alleles = np.asarray(n_alleles)
    p = np.broadcast_to(np.asarray(p, dtype=np.float64), calls.shape)
    A = int(np.max(n_alleles))
    alleles = np.arange(A)
    onehot = calls[..., None] == alleles
    new = ((1.0 - p) / error_factor)[..., None] * ~onehot
    new = np.where(onehot, p[..., None], new)
    new[calls < 0] = np.nan
    new[..., n_alleles[..., None] <= alleles] = 0.0
    return new


def synth_units(n_units, ploidy=4, n_pos=8, n_reads=200, n_alleles=2, first_unit=0, seed=20260101,
                window=(4, 8), qual=(20, 40), dedup=False):
    """Config #2 shape by default: tetraploid, 8 biallelic SNVs, 200 distinct read rows (phred path).

    Returns (reads [U, R, M, A] float64, calls [U, R, M] int8, truth [U, K, M] int8).  With dedup=True the
    phred-ignored encoding (p = 1 - error) is used instead and rows are NOT merged here (see dedup_unit)."""
    A = n_alleles
    reads = np.empty((n_units, n_reads, n_pos, A), dtype=np.float64)
    calls_all = np.empty((n_units, n_reads, n_pos), dtype=np.int8)
    truth = np.empty((n_units, ploidy, n_pos), dtype=np.int8)
    na = np.full(n_pos, A)
    for u in range(n_units):
        rng = np.random.default_rng([seed, first_unit + u])
        while True:
            haps = rng.integers(0, A, size=(ploidy, n_pos)).astype(np.int8)
            if len(np.unique(haps, axis=0)) >= min(3, ploidy, A ** n_pos):
                break
        src = haps[rng.integers(0, ploidy, size=n_reads)]
        if dedup:
            p = np.full(src.shape, 1.0 - PFEIFFER_ERROR)
        else:
            q = rng.integers(qual[0], qual[1] + 1, size=src.shape)
            p = (1.0 - PFEIFFER_ERROR) * (1.0 - 10.0 ** (-q / 10.0))
        flip = rng.random(src.shape) >= p
        calls = np.where(flip, (src + rng.integers(1, A, size=src.shape)) % A if A > 1 else src, src).astype(np.int8)
        lo, hi = window
        wlen = rng.integers(min(lo, n_pos), min(hi, n_pos) + 1, size=n_reads)
        start = (rng.random(n_reads) * (n_pos - wlen + 1)).astype(int)
        pos = np.arange(n_pos)[None, :]
        gap = (pos < start[:, None]) | (pos >= (start + wlen)[:, None])
        calls[gap] = -1
        reads[u] = as_probabilistic(calls, na, p)
        calls_all[u] = calls
        truth[u] = haps
    return reads, calls_all, truth


def dedup_unit(reads):
    """De-duplicate identical read rows (reference application/baseclass.py:207): rows in order of first
    appearance plus their counts."""
    reads = np.asarray(reads)
    if len(reads) == 0 or reads[0].size == 0:
        # no reads, or a locus without positions: every row is the same (empty) row
        n = len(reads)
        return reads[: min(n, 1)], np.full(min(n, 1), n, dtype=np.int64)
    flat = np.ascontiguousarray(reads).reshape(len(reads), -1)
    keys = flat.view("V%d" % (flat.shape[1] * flat.dtype.itemsize)).reshape(-1)  # (rows of any item type: float64 tensors, int8 calls)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    counts = np.bincount(rank[inv.reshape(-1)], minlength=len(order)).astype(np.int64)
    return reads[first[order]], counts


# ---- synthetic alignment files (test and benchmark inputs of the programs: BGZF-compressed BAM + its .bai index) ----
def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def write_bam(path, contigs, read_groups, records, index=True):
    """Write a coordinate-sorted BAM file (BGZF) and, with index=True, its `.bai`.
    contigs: [(name, length)]; read_groups: {id: sample}; records: iterable of dicts with qname, flag, ref (contig index),
    pos (0-based), mapq, cigar [(length, op char)], seq (str), qual (sequence of ints), rg (id), optionally tags_before (raw aux bytes).  Every record lies in one
    BGZF block (a new block starts when the next record would not fit), so its virtual offsets are (block << 16 | offset)."""
    import struct
    import zlib

    seq_code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    ops = {c: i for i, c in enumerate("MIDNSHP=X")}
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + \
        "".join("@RG\tID:%s\tSM:%s\n" % kv for kv in read_groups.items())
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for name, length in contigs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    out = bytearray()
    cur = bytearray(head)

    def flush():
        nonlocal cur
        if not cur:
            return
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = comp.compress(bytes(cur)) + comp.flush()
        bsize = 18 + len(body) + 8
        out.extend(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + body +
                   struct.pack("<II", zlib.crc32(bytes(cur)) & 0xFFFFFFFF, len(cur)))
        cur = bytearray()

    flush()  # the header in blocks of its own
    n_ref = len(contigs)
    bins = [dict() for _ in range(n_ref)]
    linear = [dict() for _ in range(n_ref)]
    for r in records:
        seq, qual = r["seq"], bytes(r["qual"])
        name = r["qname"].encode() + b"\0"
        cig = b"".join(struct.pack("<I", (l << 4) | ops[o]) for l, o in r["cigar"])
        ref_len = sum(l for l, o in r["cigar"] if o in "MDN=X")
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i >> 1] |= seq_code[c] << (4 if i % 2 == 0 else 0)
        tags = r.get("tags_before", b"") + b"RGZ" + r["rg"].encode() + b"\0"  # (tags_before: raw aux bytes ahead of RG, for tests)
        if "tags" in r:
            tags = r["tags"]  # (the whole aux block as given, malformed ones included: tests)
        end = r["pos"] + max(ref_len, 1)
        b = _reg2bin(r["pos"], end)
        body = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), r["mapq"], b, len(r["cigar"]), r["flag"], len(seq), -1, -1, 0) + \
            name + cig + bytes(packed) + qual + tags
        rec = struct.pack("<i", len(body)) + body
        if len(cur) + len(rec) > 60000:
            flush()
        v0 = (len(out) << 16) | len(cur)
        cur.extend(rec)
        v1 = (len(out) << 16) | len(cur)
        ch = bins[r["ref"]].setdefault(b, [])
        if ch and ch[-1][1] == v0:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        for w in range(r["pos"] >> 14, ((end - 1) >> 14) + 1):
            linear[r["ref"]].setdefault(w, v0)
    flush()
    out.extend(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the BGZF end-of-file block
    with open(path, "wb") as f:
        f.write(out)
    if index:
        idx = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
        for t in range(n_ref):
            idx += struct.pack("<i", len(bins[t]))
            for b, chunks in bins[t].items():
                idx += struct.pack("<Ii", b, len(chunks))
                for v0, v1 in chunks:
                    idx += struct.pack("<QQ", v0, v1)
            n_intv = (max(linear[t]) + 1) if linear[t] else 0
            idx += struct.pack("<i", n_intv)
            last = 0
            for w in range(n_intv):
                last = linear[t].get(w, last)
                idx += struct.pack("<Q", last)
        with open(path + ".bai", "wb") as f:
            f.write(idx)


def synth_assembly_inputs(directory, n_loci=200, n_samples=2, reads_per_locus=60, ploidy=4, n_snvs=8, locus_len=120, seed=20260102, gap=80):
    """A synthetic `mchap assemble` job on disk: one contig, n_loci target windows with n_snvs SNVs each (BED4 + VCF +
    FASTA), and one coordinate-sorted, indexed BAM per sample with paired-looking reads drawn from ploidy haplotypes per
    (locus, sample).  Returns dict(bams, bed, vcf, fasta)."""
    import os

    rng = np.random.default_rng(seed)
    length = n_loci * (locus_len + gap) + gap
    refseq = rng.choice(list("ACGT"), size=length)
    bed_lines, vcf_lines, snv_pos = [], [], []
    alt_of = {"A": "C", "C": "T", "G": "A", "T": "G"}
    for li in range(n_loci):
        start = gap + li * (locus_len + gap)
        pos = np.sort(rng.choice(np.arange(start + 5, start + locus_len - 5), size=n_snvs, replace=False))
        snv_pos.append(pos)
        bed_lines.append("chrS\t%d\t%d\tlocus%05d" % (start, start + locus_len, li))
        for p in pos:
            vcf_lines.append("chrS\t%d\t.\t%s\t%s\t.\t.\t." % (p + 1, refseq[p], alt_of[refseq[p]]))
    os.makedirs(directory, exist_ok=True)
    bed, vcf, fasta = (os.path.join(directory, n) for n in ("targets.bed", "snvs.vcf", "ref.fa"))
    open(bed, "w").write("\n".join(bed_lines) + "\n")
    open(vcf, "w").write("##fileformat=VCFv4.3\n##contig=<ID=chrS,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % length + "\n".join(vcf_lines) + "\n")
    s = "".join(refseq)
    open(fasta, "w").write(">chrS\n" + "\n".join(s[i:i + 60] for i in range(0, length, 60)) + "\n")
    open(fasta + ".fai", "w").write("chrS\t%d\t6\t60\t61\n" % length)
    bams = []
    for si in range(n_samples):
        sample = "S%03d" % si
        recs = []
        for li in range(n_loci):
            start = gap + li * (locus_len + gap)
            haps = rng.integers(0, 2, size=(ploidy, n_snvs))
            for ri in range(reads_per_locus):
                h = haps[rng.integers(0, ploidy)]
                rl = int(rng.integers(60, 100))
                p0 = int(rng.integers(start - 10, start + locus_len - rl + 10))
                seq = refseq[p0:p0 + rl].copy()
                for j, p in enumerate(snv_pos[li]):
                    if p0 <= p < p0 + rl and h[j]:
                        seq[p - p0] = alt_of[refseq[p]]
                err = rng.random(rl) < 0.003
                seq[err] = rng.choice(list("ACGT"), size=int(err.sum()))
                recs.append(dict(qname="r%d_%d_%d" % (si, li, ri), flag=0, ref=0, pos=p0, mapq=60, cigar=[(rl, "M")], seq="".join(seq),
                                 qual=rng.integers(20, 41, size=rl).astype(np.uint8).tolist(), rg="rg" + sample))
        recs.sort(key=lambda r: r["pos"])
        path = os.path.join(directory, sample + ".bam")
        write_bam(path, [("chrS", length)], {"rg" + sample: sample}, recs)
        bams.append(path)
    return dict(bams=bams, bed=bed, vcf=vcf, fasta=fasta)
