"""Differential fuzz of low-quality units (chains that never settle; mostly at most 64 reads: requests evaluated side by side
from the code table in LDS, up to 24 K rows; the resumed chains of the phased sampler keep decision contexts per genotype) of
random shapes against the oracle, the phased and the speculative sampler.
Needs a GPU; test infrastructure, like tests/fuzz_kernels.py.
    python tests/fuzz_moving.py [cases] [first_seed]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np

from oracle import binding as orc


def run(cases=40, seed0=9000):
    from mchap_amd import DenovoMCMC
    from mchap_amd.assemble import break_table
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    bad = 0
    for case in range(cases):
        rng = np.random.default_rng([seed0, case])
        K = int(rng.choice([2, 3, 4, 4, 4, 5, 6]))
        A = int(rng.choice([2, 2, 2, 3]))
        bits = 1 if A == 2 else 2
        M = int(rng.integers(5, min(24, 44, 192 // K, 64 // bits) + 1))
        R = int(rng.integers(4, 65)) if rng.random() < 0.75 else int(rng.integers(65, 300))  # (round 5: now and then beyond one read chunk)
        F = [None, 0.0, 0.2][int(rng.integers(0, 3))]
        steps = int(rng.integers(150, 450)) if rng.random() < 0.6 else int(rng.integers(450, 1400))  # (... and long enough for the decision
        # contexts of the resumed chains to be replaced, paused and tried again)
        units = int(rng.integers(2, 7))
        first = int(rng.integers(0, 10 ** 6))
        reads, _, _ = synth_units(units, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, qual=(3, int(rng.integers(8, 25))), window=(min(4, M), M),
                                  first_unit=first)
        kw = dict(ploidy=K, n_alleles=[A] * M, inbreeding=F, steps=steps, chains=2, random_seed=int(rng.integers(1, 10 ** 6)))
        for kernel in (5, 3):
            os.environ["MCHAP_HIP_KERNEL"] = str(kernel)
            model = DenovoMCMC(**kw)
            got = model.fit_batch(list(reads))
            ok = True
            for u, tr in enumerate(got):
                cfg = orc.make_cfg(K, steps, 2, F, (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=kw["random_seed"], stream_id=u,
                                   break_table=break_table(M, 1.0, 3.0))
                g, l, code = orc.denovo_fit(cfg, reads[u], [A] * M, None)
                if code != 0 or not np.array_equal(tr.genotypes, sort_haplotypes(g)) or not np.allclose(tr.llks, l, rtol=1e-10, atol=1e-9):
                    ok = False
            bad += 0 if ok else 1
            print("case %d kernel %d K=%d M=%d A=%d R=%d F=%s steps=%d units=%d: %s" % (case, kernel, K, M, A, R, F, steps, units, "ok" if ok else "FAIL"),
                  flush=True)
    print("SOAK-MOVING FAILURES: %d" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 9000) else 0)
