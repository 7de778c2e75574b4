"""Differential fuzz: random shapes / sampler settings, every sampler kernel against the CPU oracle.

    python tests/fuzz_kernels.py [n_cases] [seed]        (needs a GPU; test infrastructure, like tests/)
"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np

from mchap_amd import DenovoMCMC
from mchap_amd.classes import sort_haplotypes
from mchap_amd.synth import synth_units
from test_gpu_denovo import _oracle_trace


def run(n_cases, seed, ploidies=(2, 3, 4, 5, 6, 8), read_depths=(1, 5, 30, 64, 70, 130, 200, 300), tempering=None, max_pos=12):
    """ploidies / read_depths: what the cases draw from; tempering: None = random, True = always a ladder, False = never."""
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        K = int(rng.choice(ploidies))
        M = int(rng.integers(1, max_pos + 1))
        A = int(rng.choice([2, 2, 3, 4]))
        R = int(rng.choice(read_depths))
        chains = int(rng.integers(1, 4))
        U = int(rng.integers(1, 4))
        temps = [(1.0,), (1.0,), (0.3, 1.0), (0.2, 0.6, 1.0)][int(rng.integers(0, 4))]
        if tempering:
            temps = [(0.3, 1.0), (0.2, 0.6, 1.0), (0.001, 0.01, 0.1, 1.0)][int(rng.integers(0, 3))]
        elif tempering is False:
            temps = (1.0,)
        F = [None, None, 0.0, 0.15][int(rng.integers(0, 4))]
        pr = [float(rng.choice([-1.0, 0.3, 0.5, 1.0])) for _ in range(3)]
        if rng.random() < 0.3:
            pr = [0.5, 0.5, 1.0]
        steps = int(rng.choice([30, 80, 150]))
        lo = int(rng.integers(1, M + 1))
        reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=1000 + case,
                                  window=(lo, M), qual=(5, 40))
        kw = dict(ploidy=K, n_alleles=[A] * M, inbreeding=F, steps=steps, chains=chains, temperatures=temps,
                  random_seed=int(rng.integers(0, 2 ** 31)), recombination_step_probability=pr[0],
                  partial_dosage_step_probability=pr[1], dosage_step_probability=pr[2],
                  llk_cache_threshold=int(rng.choice([-1, 100])))
        m0 = DenovoMCMC(kernel=2, **kw)
        ref = []
        for u in range(U):
            g, l = _oracle_trace(m0, reads[u], m0.n_alleles, None, u)
            ref.append((sort_haplotypes(g), l))
        res = []
        for k in (1, 2, 3, 4, 5, 6):
            # kernels 1 and 4 live in the parity suite's library (MCHAP_HIP_TEST_KERNELS); "6" = the phased sampler with the
            # lane-per-request table completion of that library (tuning flag 64)
            test_lib = k in (1, 4, 6)
            if test_lib:
                os.environ["MCHAP_HIP_TEST_KERNELS"] = "1"
            if k == 6:
                os.environ["MCHAP_HIP_FLAGS"] = "64"
            try:
                tr = DenovoMCMC(kernel=5 if k == 6 else k, **kw).fit_batch(list(reads))
            except NotImplementedError:
                res.append("n/a")
                continue
            finally:
                os.environ.pop("MCHAP_HIP_TEST_KERNELS", None)
                os.environ.pop("MCHAP_HIP_FLAGS", None)
            ok = all(np.array_equal(tr[u].genotypes, ref[u][0]) and
                     np.allclose(tr[u].llks, ref[u][1], rtol=1e-10, atol=1e-9, equal_nan=True) for u in range(U))
            res.append("ok" if ok else "FAIL")
            bad += not ok
        print("case %3d K=%d M=%2d A=%d R=%3d C=%d U=%d T=%d F=%s p=%s steps=%d cache=%d : %s" % (
            case, K, M, A, R, chains, U, len(temps), F, pr, steps, kw["llk_cache_threshold"], " ".join(res)), flush=True)
    print("FAILURES: %d" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
