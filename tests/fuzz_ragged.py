"""Differential fuzz of RAGGED launches: batches whose units differ in SNV count, alleles per SNV, read depth, base quality,
inbreeding and stream -- what `mchap assemble` hands the library for a real VCF (DenovoRaggedBatch: one launch per ploidy) --
at batch sizes from one unit to a few hundred (the phased sampler picks its launch structure from the number of chains:
parts per chain of the table completion, resume rounds, list sizes).  Every step of every chain of every unit against the
CPU oracle on the same Philox streams; the device-side posterior summary against the host classes on the oracle's trace.

    python tests/fuzz_ragged.py [n_cases] [seed] [max_units] [max_pos]        (needs a GPU; test infrastructure, like tests/)
"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np

from oracle import binding as orc


def random_units(rng, K, U, max_pos, deep=600):
    from mchap_amd.encoding import unique_counts
    from mchap_amd.synth import synth_units

    units = []
    shapes = [(int(rng.integers(1, max_pos + 1)), int(rng.choice([2, 2, 2, 3, 4]))) for _ in range(int(rng.integers(1, 9)))]
    for i in range(U):
        M, A = shapes[int(rng.integers(0, len(shapes)))]
        M = min(M, 32) if A > 2 else M  # (the library's packed haplotype: 64 bits at two bits per tri- / tetra-allelic position)
        R = int(rng.choice([1, 3, 8, 20, 40, 70, 130, 260, deep]))
        qual = [(20, 40), (20, 40), (3, 20), (5, 15), (30, 30)][int(rng.integers(0, 5))]
        lo = int(rng.integers(1, M + 1))
        rd, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=int(rng.integers(0, 1 << 30)), window=(lo, M), qual=qual)
        rd, counts = rd[0], None
        if rng.random() < 0.4:  # the program's default encoding: distinct rows with counts
            rd, counts = unique_counts(rd)
        # per-position allele counts: some positions of a tri/tetra-allelic unit are biallelic
        n_alleles = [int(rng.integers(2, A + 1)) for _ in range(M)] if A > 2 else [2] * M
        n_alleles[int(rng.integers(0, M))] = A
        for j, n in enumerate(n_alleles):
            if n < A:
                # probability mass of the alleles a position does not have goes nowhere (encoded as 0): as the encoder writes it
                rd[:, j, n:] = 0.0
        units.append(dict(reads=rd, counts=counts, n_alleles=n_alleles, ploidy=K, inbreeding=[None, None, 0.1, 0.3][int(rng.integers(0, 4))],
                          stream_id=int(rng.integers(0, 4))))
    return units


def oracle_unit(u, steps, chains, seed, burn, threshold):
    from mchap_amd import GenotypeMultiTrace
    from mchap_amd.classes import sort_haplotypes
    from tests.helpers import beta_break_table

    M = u["reads"].shape[1]
    cfg = orc.make_cfg(u["ploidy"], steps, chains, u["inbreeding"], (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=seed,
                       stream_id=u["stream_id"], break_table=beta_break_table(M, 1.0, 3.0))
    g, l, code = orc.denovo_fit(cfg, u["reads"], list(u["n_alleles"]), u["counts"])
    assert code == 0
    g = sort_haplotypes(g)
    tr = GenotypeMultiTrace._from_sorted(g, l).burn(burn)
    post = tr.posterior()
    sup = post.mode_genotype_support()
    mg, gp = sup.mode_genotype()
    return g, l, dict(spm=float(sup.probabilities.sum()), gpm=float(gp), mode_genotype=mg, mci=int(tr.replicate_incongruence(threshold)),
                      n=len(post.probabilities))


def run(n_cases, seed, max_units=300, max_pos=24, verbose=True):
    import torch

    from mchap_amd import DenovoMCMC
    from mchap_amd.assemble import unpack_trace
    from mchap_amd.device import DenovoRaggedBatch

    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        K = int(rng.choice([2, 3, 4, 4, 4, 6]))
        U = int(np.exp(rng.uniform(0, np.log(max_units))))
        steps = int(rng.choice([60, 150, 300, 300, 700]))
        chains = int(rng.integers(1, 4))
        burn = steps // 2
        sd = int(rng.integers(0, 2 ** 31))
        t0 = time.time()
        units = random_units(rng, K, U, max_pos if K <= 4 else 16, deep=600 if K <= 4 else 260)
        model = DenovoMCMC(ploidy=K, n_alleles=[2], steps=steps, chains=chains, random_seed=sd)
        b = DenovoRaggedBatch(model, units)
        b.run(burn)
        torch.cuda.synchronize()
        res = b.results()
        trace = b.d_trace.cpu().numpy().view(np.uint64)
        llks = b.d_llks.cpu().numpy()
        fixed = b.d_fixed.cpu().numpy()
        n_bad = 0
        for i, u in enumerate(units):
            D = b.units_host[i]
            M, A = int(D["n_pos"]), int(D["max_allele"])
            g, l, summary = oracle_unit(u, steps, chains, sd, burn, 0.6)
            wph = b.wph  # (2: the general sampler's 128-bit haplotype words -- a batch whose widest unit x most alleles exceed 64 bits)
            w = trace[int(D["trace_off"]): int(D["trace_off"]) + chains * steps * K * wph].reshape((chains, steps, K) + ((wph,) if wph > 1 else ()))
            got = unpack_trace(w, fixed[int(D["fixed_off"]): int(D["fixed_off"]) + M], A, wph)
            lk = llks[int(D["llk_off"]): int(D["llk_off"]) + chains * steps].reshape(chains, steps)
            ok = np.array_equal(got, g) and np.allclose(lk, l, rtol=1e-10, atol=1e-9, equal_nan=True)
            r = res[i]
            ok = ok and np.array_equal(r["mode_genotype"], summary["mode_genotype"]) and abs(r["gpm"] - summary["gpm"]) < 1e-12 \
                and abs(r["spm"] - summary["spm"]) < 1e-12 and r["mci"] == summary["mci"] and len(r["probabilities"]) == summary["n"]
            if not ok:
                n_bad += 1
                if verbose:
                    first = np.argwhere((got != g).any(axis=(2, 3)))
                    print("   unit %d: M=%d A=%d R=%d counts=%s F=%s: first differing (chain, step) %s" % (
                        i, M, A, u["reads"].shape[0], u["counts"] is not None, u["inbreeding"], first[0].tolist() if len(first) else "summary only"), flush=True)
        bad += n_bad
        if verbose:
            print("case %3d K=%d units=%3d shapes=%d steps=%3d chains=%d %-58s %s  (%.0f s)" % (
                case, K, U, len({(x["reads"].shape[1], x["reads"].shape[2]) for x in units}), steps, chains, b_name(b), "ok" if n_bad == 0 else "FAIL x %d" % n_bad,
                time.time() - t0), flush=True)
    print("FAILURES: %d" % bad)
    return bad


def b_name(batch):
    from mchap_amd import _lib

    return _lib.sampler_name(batch.cfg, batch.units_host)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    sys.exit(1 if run(a[0] if len(a) > 0 else 20, a[1] if len(a) > 1 else 1, a[2] if len(a) > 2 else 300, a[3] if len(a) > 3 else 24) else 0)
