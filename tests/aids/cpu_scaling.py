"""Host-side check (not part of the product): scaling of the CPU oracle baseline with OpenMP threads."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as orc
from mchap_amd.assemble import break_table
from mchap_amd.synth import synth_units
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reads, _, _ = synth_units(n)
for thr in (1, 8, 32, 64, 128, 256):
    if thr > (os.cpu_count() or 1):
        break
    cfg = orc.make_cfg(4, 1000, 2, None, (1.0,), llk_cache_threshold=100, seed=42, break_table=break_table(8, 1.0, 3.0))
    m = min(n, max(64, thr * 16))
    t = time.perf_counter()
    orc.denovo_fit_batch(cfg, reads[:m], [2] * 8, n_threads=thr, keep_traces=False)
    dt = time.perf_counter() - t
    print("threads %3d: %5d loci in %6.2f s -> %8.1f loci/s (%.1f per thread)" % (thr, m, dt, m / dt, m / dt / thr), flush=True)
