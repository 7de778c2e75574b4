"""CPU analysis (not a test): what a chain that keeps moving does at SUB-STEP level, from the oracle built with -DORC_MOVE_LOG
(oracle/_build/libmchap_oracle_movelog.so: every accepted mutation / structural move and every visited interval step logged as
a hash of the ordered genotype).  Sizes the per-genotype decision contexts of the speculative sampler:
distinct ordered genotypes per chain, hit rates of an LRU of N contexts, structural moves per step, distinct (genotype, interval)
pairs.

  python tests/aids/analyze_moves.py example [locus015,locus012,...]     docs/example units (the pileup fixture)
  python tests/aids/analyze_moves.py moving 16|40 [units]                 bench.py's extra.moving shapes
"""
import ctypes as C
import os
import subprocess
import sys
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as orc  # noqa: E402

SO = os.path.join(ROOT, "oracle", "_build", "libmchap_oracle_movelog.so")


def movelog_lib():
    src = os.path.join(ROOT, "oracle", "mchap_oracle.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-DORC_MOVE_LOG",
                               "-shared", "-o", SO, src, "-lm"])
    lib = C.CDLL(SO)
    lib.orc_move_log_count.restype = C.c_size_t
    return lib


def run_unit(lib, cfg, reads, n_alleles, counts, cap=1 << 24):
    reads = np.ascontiguousarray(reads, np.float64)
    R, M, A = reads.shape
    na = np.ascontiguousarray(n_alleles, np.int8)
    rc = None if counts is None else np.ascontiguousarray(counts, np.int64)
    g = np.zeros((cfg.chains, cfg.steps, cfg.ploidy, M), np.int8)
    l = np.zeros((cfg.chains, cfg.steps))
    buf = np.zeros(cap, np.uint64)
    lib.orc_move_log_set(buf.ctypes.data_as(C.c_void_p), C.c_size_t(cap))
    code = lib.orc_denovo_fit(C.byref(cfg), reads.ctypes.data_as(C.c_void_p), R, M, A,
                              None if rc is None else rc.ctypes.data_as(C.c_void_p), na.ctypes.data_as(C.c_void_p), None,
                              g.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), None)
    n = lib.orc_move_log_count()
    lib.orc_move_log_set(None, C.c_size_t(0))
    assert code == 0 and n <= cap, (code, n)
    return buf[:n]


def lru_hits(seq, size):
    d = OrderedDict()
    hits = 0
    for k in seq:
        if k in d:
            d.move_to_end(k)
            hits += 1
        else:
            d[k] = 1
            if len(d) > size:
                d.popitem(last=False)
    return hits


def analyse(log, steps, chains):
    kind = (log & np.uint64(3)).astype(np.int64)
    key = log >> np.uint64(2)
    n_steps = int((kind == 0).sum())
    # split into chains: step starts are in order; chain c = steps [c*steps, (c+1)*steps)
    starts = np.flatnonzero(kind == 0)
    out = []
    for c in range(n_steps // steps):
        lo = starts[c * steps]
        hi = starts[(c + 1) * steps] if (c + 1) * steps < len(starts) else len(log)
        k, h = kind[lo:hi], key[lo:hi]
        mut = h[k == 1]
        st = h[k == 2]
        moves = h[(k == 1) | (k == 2)]
        ivis = h[k == 3]
        row = dict(mut_moves_per_step=len(mut) / steps, struct_moves_per_step=len(st) / steps,
                   interval_visits_per_step=len(ivis) / steps, distinct_genotypes=len(set(moves.tolist())),
                   distinct_interval_keys=len(set(ivis.tolist())))
        for size in (4, 8, 16, 32, 64, 128, 256):
            row["lru%d" % size] = lru_hits(moves.tolist(), size) / max(1, len(moves))
        # distinct genotypes in which the chain STANDS at the start of a mutation step / at an interval visit
        out.append(row)
    return out


def main():
    lib = movelog_lib()
    mode = sys.argv[1]
    rows = []
    if mode == "example":
        from mchap_amd import application
        from tests.helpers import beta_break_table

        want = sys.argv[2].split(",") if len(sys.argv) > 2 else None
        steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
        samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(ROOT, "tests", "golden", "example_biparental.npz"))
        captured = []

        def sampler(units, settings):
            captured.extend(units)
            raise StopIteration

        class _N:
            known = False

            def __getitem__(self, sl):
                return "N" * (sl.stop - sl.start)

        tg = [t for t in targets if want is None or t[3] in want] if want else targets
        try:
            list(application.assemble(None, variants, {"chr1": _N()}, application.MatrixSource(samples, matrices), ploidy=4, steps=steps,
                                      burn=steps // 2, chains=2, seed=42, targets=tg, report=(), sampler=sampler))
        except (StopIteration, RuntimeError):
            pass
        print("units", len(captured))
        for u in captured:
            M = u["reads"].shape[1]
            cfg = orc.make_cfg(u["ploidy"], steps, 2, u["inbreeding"], u["temps"], llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=42,
                               stream_id=u["stream_id"], break_table=beta_break_table(M, 1.0, 3.0))
            log = run_unit(lib, cfg, u["reads"], list(u["n_alleles"]), u["counts"])
            for r in analyse(log, steps, 2):
                r["M"] = M
                r["R"] = u["reads"].shape[0]
                rows.append(r)
    else:
        from mchap_amd.assemble import break_table
        from mchap_amd.synth import synth_units

        R = int(sys.argv[2])
        U = int(sys.argv[3]) if len(sys.argv) > 3 else 16
        steps = 1000
        reads, _, _ = synth_units(U, ploidy=4, n_pos=8, n_reads=R, qual=(3, 20))
        for i in range(U):
            cfg = orc.make_cfg(4, steps, 2, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX, stream_id=i,
                               break_table=break_table(8, 1.0, 3.0))
            log = run_unit(lib, cfg, reads[i], [2] * 8, None)
            for r in analyse(log, steps, 2):
                r["M"] = 8
                r["R"] = R
                rows.append(r)
    keys = ["M", "R", "mut_moves_per_step", "struct_moves_per_step", "interval_visits_per_step", "distinct_genotypes", "distinct_interval_keys",
            "lru4", "lru8", "lru16", "lru32", "lru64", "lru128", "lru256"]
    print(" ".join("%10s" % k[:10] for k in keys))
    rows.sort(key=lambda r: -r["mut_moves_per_step"])
    for r in rows[:60]:
        print(" ".join(("%10.3f" % r[k]) if isinstance(r[k], float) else ("%10d" % r[k]) for k in keys))


if __name__ == "__main__":
    main()
