"""Differential soak of the `mchap call` sampler (call_mcmc_kernel + call_coast_kernel) against the oracle on random shapes:
ploidy 1-8 (and 9-12 without the Gibbs memo), 2-70 known haplotypes, 1-400 reads of mixed quality, 1-5 chains, both step types,
with and without a prior / frequencies / read counts / an initial genotype, at random lanes per wavefront of the coast kernel.
Needs a GPU; uses the oracle, hence lives under tests/ (tools/ never imports it).  python tests/fuzz_call.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n_cases, seed, only=None, lanes_override=None):
    from oracle import binding as orc

    from mchap_amd.calling_mcmc import CallingMCMC
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        K = int(rng.choice([1, 2, 3, 4, 4, 5, 6, 8, 9, 12]))
        M = int(rng.integers(2, 11))
        H = int(min(rng.choice([2, 3, 5, 8, 16, 17, 33, 70]), 2 ** M))
        R = int(rng.choice([1, 3, 12, 40, 64, 130, 400]))
        chains = int(rng.integers(1, 6))
        steps = int(rng.choice([30, 120, 500]))
        step_type = "Gibbs" if rng.random() < 0.8 else "Metropolis-Hastings"
        qual = (2, 8) if rng.random() < 0.4 else (5, 30)
        U = 3
        # (a few reads cover every position: a position no read covers makes pairs of known haplotypes equally likely, and which of
        # two sums that differ in their last bit is the larger is not something two implementations agree on -- seed 24, case 101
        # of the first version of this script: the reference's own greedy_caller sided with the GPU there, not with the oracle)
        reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=int(rng.integers(0, 10 ** 6)), window=(M, M) if R <= 12 else (1, M), qual=qual)
        haps = np.zeros((U, H, M), np.int8)
        for u in range(U):
            pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(8 * H, M)).astype(np.int8)]), axis=0)
            rng.shuffle(pool)
            while len(pool) < H:
                pool = np.concatenate([pool, pool])  # (duplicates among the known haplotypes are allowed)
            haps[u] = pool[:H]
        counts = rng.integers(1, 4, size=(U, R)).astype(np.int64) if rng.random() < 0.4 else None
        F = rng.choice([0.0, 0.1, 0.4], size=U)
        fr = rng.dirichlet(np.ones(H), size=U) if rng.random() < 0.3 else None
        prior = None if rng.random() < 0.3 else (F, fr)
        ini = np.sort(rng.integers(0, H, size=(U, K)), axis=1) if rng.random() < 0.3 else None
        lanes = str(rng.choice(["", "0", "1", "3", "7", "15"]))
        s_ = int(rng.integers(0, 10 ** 6))
        if only is not None and case != only:
            continue
        if lanes_override is not None:
            lanes = lanes_override
        if lanes:
            os.environ["MCHAP_HIP_CALL_LANES"] = lanes
        else:
            os.environ.pop("MCHAP_HIP_CALL_LANES", None)
        model = CallingMCMC(ploidy=K, haplotypes=haps[0], prior=None, steps=steps, chains=chains, random_seed=s_, step_type=step_type)
        traces = model.fit_batch(reads, counts, haplotypes=haps, prior=prior, initial=ini)
        for u in range(U):
            pr = None if prior is None else (float(F[u]), None if fr is None else fr[u])
            g, l = orc.call_mcmc(reads[u], haps[u], K, steps=steps, chains=chains, step_type=0 if step_type == "Gibbs" else 1,
                                 read_counts=None if counts is None else counts[u], prior=pr, initial=None if ini is None else ini[u],
                                 rng_kind=orc.RNG_PHILOX, seed=s_, stream_id=u)
            ok = np.array_equal(traces[u].genotypes, g) and np.allclose(traces[u].llks, l, rtol=1e-10, atol=1e-9)
            if not ok:
                bad += 1
                if only is not None:  # (a single case: say where)
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez("gpurun_out/fuzz_call_case.npz", reads=reads[u], haps=haps[u], K=K, F=F[u], steps=steps, chains=chains, seed=s_, unit=u,
                             gpu_g=traces[u].genotypes, gpu_l=traces[u].llks, orc_g=g, orc_l=l)
                    print("  F", F[u], "freqs", fr is not None, "haplotypes of the first genotypes:")
                    for a in sorted(set(traces[u].genotypes[0, 0].tolist()) | set(g[0, 0].tolist())):
                        print("   ", a, haps[u][a].tolist(), "same as", [int(b) for b in range(H) if np.array_equal(haps[u][b], haps[u][a])])
                    dg = np.argwhere((traces[u].genotypes != g).any(axis=-1))
                    dl = np.argwhere(~np.isclose(traces[u].llks, l, rtol=1e-10, atol=1e-9))
                    print("  first genotype difference (chain, step):", dg[:3].tolist(), "first llk difference:", dl[:3].tolist())
                    for c, st in dg[:2].tolist():
                        print("  gpu", traces[u].genotypes[c, max(st - 1, 0): st + 2].tolist(), traces[u].llks[c, max(st - 1, 0): st + 2].tolist())
                        print("  orc", g[c, max(st - 1, 0): st + 2].tolist(), l[c, max(st - 1, 0): st + 2].tolist())
                    for c, st in dl[:2].tolist():
                        print("  llk at", c, st, repr(float(traces[u].llks[c, st])), repr(float(l[c, st])), traces[u].genotypes[c, st].tolist(), g[c, st].tolist())
                print("DIFFERENCE case %d unit %d: K %d H %d M %d R %d chains %d steps %d %s lanes %r prior %s counts %s initial %s seed %d"
                      % (case, u, K, H, M, R, chains, steps, step_type, lanes, prior is not None, counts is not None, ini is not None, s_), flush=True)
    os.environ.pop("MCHAP_HIP_CALL_LANES", None)
    print("fuzz_call: %d cases x 3 units, seed %d: %d differences" % (n_cases, seed, bad), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None  # (one case of the sequence alone, optionally at other lanes: argv[4])
    sys.exit(1 if run(n, sd, only, sys.argv[4] if len(sys.argv) > 4 else None) else 0)
