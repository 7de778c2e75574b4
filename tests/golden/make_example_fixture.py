#!/usr/bin/env python3
"""Generate tests/golden/example_biparental.npz: BASELINE.json configs[0] (`mchap assemble` on docs/example) as data.

For every target of docs/example/input/bed/targets20.bed and every one of the 22 samples of docs/example/input/bam/ the
read-by-SNV character matrix and the summed base qualities that extract_read_variants (io/bam.py:54-229) yields -- read
from the BAM files with mchap_amd.io (whose reader is pinned to the reference's own matrices by
tests/test_io_bam_matrices.py) -- plus the targets, the SNVs of input/vcf/snvs.vcf.gz inside them and the contig length of
input/fasta/chr1.fa.gz.fai (the FASTA itself is not part of the reference's repository: reference bases that are not an
SNV are N).  Also the sample columns of the record lines printed in docs/example/bi-parental.ipynb (the real reference's
output for targets4.bed: numba's own random stream, so they are compared statistically).  Inputs and expected outputs
only: data, no reference source.  Build container only.  Usage: python tests/golden/make_example_fixture.py"""
import glob
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
EX = "/root/reference/docs/example"

from mchap_amd import io  # noqa: E402

paths = sorted(glob.glob(os.path.join(EX, "input/bam/*.bam")))
table = io.sample_bam_table(paths)
samples = list(table)
bams = {s: io.read_alignments(p) for s, p in table.items()}
targets = io.read_bed4(os.path.join(EX, "input/bed/targets20.bed"))
_, variants = io.read_vcf(os.path.join(EX, "input/vcf/snvs.vcf.gz"))
ref = io.Reference(os.path.join(EX, "input/fasta/chr1.fa.gz"), allow_index_only=True)
out = dict(samples=np.array(samples), contigs=np.array([c for c, _ in ref.contigs]), contig_lengths=np.array([n for _, n in ref.contigs]),
           target_contig=np.array([t[0] for t in targets]), target_start=np.array([t[1] for t in targets]),
           target_stop=np.array([t[2] for t in targets]), target_name=np.array([t[3] for t in targets]))
n_reads = 0
for li, (contig, start, stop, name) in enumerate(targets):
    locus = io.DenovoLocus(contig, start, stop, name, variants, ref.fetch(contig, start, stop), sequence_known=ref.known)
    out["pos_%d" % li] = np.array(locus.positions, dtype=np.int64)
    out["alleles_%d" % li] = np.array(["".join(a) for a in locus.alleles])  # e.g. "AC": REF then ALTs, one character each
    for si, s in enumerate(samples):
        chars, quals = io.extract_read_variants(locus, bams[s], s)
        out["chars_%d_%d" % (li, si)] = np.frombuffer("".join("".join(r) for r in chars).encode("ascii"), dtype=np.uint8).reshape(chars.shape)
        out["quals_%d_%d" % (li, si)] = quals.astype(np.int16)
        n_reads += len(chars)
np.savez_compressed(os.path.join(HERE, "example_biparental.npz"), **out)
print("targets", len(targets), "samples", len(samples), "read rows", n_reads,
      "bytes", os.path.getsize(os.path.join(HERE, "example_biparental.npz")))

# the notebook's printed records (targets4.bed, default settings: ploidy 4, seed 42, 2000 steps): FILTER, INFO and the
# GT / GPM of every sample
nb = json.load(open(os.path.join(EX, "bi-parental.ipynb")))
ansi = re.compile(r"\x1b\[[0-9;]*[mK]")
recs = {}
for cell in nb["cells"]:
    src = "".join(cell.get("source", []))
    if cell["cell_type"] != "code" or not src.startswith("zcat assemble.vcf.gz | grep \"locus"):
        continue
    for o in cell.get("outputs", []):
        for line in "".join(o.get("text", [])).splitlines():
            f = ansi.sub("", line).split("\t")
            if len(f) > 9:
                recs[f[2]] = dict(pos=int(f[1]), ref=f[3], alts=f[4].split(","), filter=f[6], info=f[7], format=f[8], samples=f[9:])
json.dump(recs, open(os.path.join(HERE, "example_notebook_records.json"), "w"), indent=0)
print("notebook records:", sorted(recs))
