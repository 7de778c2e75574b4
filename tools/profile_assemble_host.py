"""Measurement aid: cProfile of application.assemble on the docs/example pileups x 16 (7 040 units): where the host time of
the program goes around the sampler launch."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mchap_amd import application

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(root, "tests", "golden", "example_biparental.npz"))


class NSeq:
    known = False  # (application.assemble: the variants' REF alleles are written at their positions)

    def __getitem__(self, sl):
        return "N" * (sl.stop - sl.start)


def once(rep, tm, upb=None):
    source = application.MatrixSource(samples, matrices)
    return list(application.assemble(None, variants, {c: NSeq() for c, _ in contigs}, source, ploidy=4, steps=2000, burn=1000, chains=2,
                                     seed=42, targets=list(targets) * rep, timings=tm, units_per_block=upb))


once(1, {})
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 16
upb = int(sys.argv[2]) if len(sys.argv) > 2 else None  # units per block (several blocks: host and device work overlap)
tm = {}
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
lines = once(rep, tm, upb)
pr.disable()
print("wall %.2f s" % (time.perf_counter() - t), {k: round(v, 3) for k, v in tm.items()})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
