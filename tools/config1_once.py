"""Measurement aid: bench.py's extra.config1 (docs/example through the program) alone: 440 units, and the same x16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mchap_amd import application

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(ROOT, "tests", "golden", "example_biparental.npz"))


class NSeq:
    known = False

    def __getitem__(self, sl):
        return "N" * (sl.stop - sl.start)


src = application.MatrixSource(samples, matrices)
reps = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 1, 16, 16)
upb = int(sys.argv[3]) if len(sys.argv) > 3 else None  # units per block (default: by free memory; several blocks overlap host and device work)
for rep in reps:
    for bp in ((None, False) if len(sys.argv) < 3 else (None,)):
        tm = {}
        t = time.perf_counter()
        lines = list(application.assemble(None, variants, {c: NSeq() for c, _ in contigs}, src, ploidy=4, steps=2000, burn=1000, chains=2, seed=42,
                                          targets=list(targets) * rep, timings=tm, block_path=bp, units_per_block=upb))
        dt = time.perf_counter() - t
        print("x%d block_path=%s upb=%s %.1f ms  %.0f units/s  %s" % (rep, bp, upb, dt * 1e3, tm["units"] / dt, {k: round(v * 1e3, 1) for k, v in tm.items() if k.endswith("_s")}), flush=True)
