"""Profiling helper (not part of the product): phase timers / event counters of the sampler on docs/example targets (the pileup
fixture), e.g. the chains that never settle.  Usage: MCHAP_HIP_LIB=mchap_amd/csrc/libmchap_hip_phases.so python tools/stats_example.py
locus015[,locus012] [samples]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import _lib
_lib.SO = os.environ.get("MCHAP_HIP_LIB") or os.path.join(_lib.CSRC, "libmchap_hip_phases.so")
from mchap_amd import application

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(ROOT, "tests", "golden", "example_biparental.npz"))
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["locus015"]
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else len(samples)
STEPS = 2000


class NSeq:
    known = False

    def __getitem__(self, sl):
        return "N" * (sl.stop - sl.start)


src = application.MatrixSource(samples[:n_samples], {k: v for k, v in matrices.items()})
tg = [t for t in targets if t[3] in names]
L = _lib.lib()
out = (C.c_ulonglong * 64)()
for rep in range(2):
    L.mchap_debug_stats(out, 1)
    tm = {}
    t = time.perf_counter()
    lines = list(application.assemble(None, variants, {c: NSeq() for c, _ in contigs}, src, ploidy=4, steps=STEPS, burn=1000, chains=2, seed=42,
                                      targets=tg, timings=tm))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
L.mchap_debug_stats(out, 0)
units = tm["units"]
ws = units * 2 * STEPS
print(names, "units", units, "wall %.1f ms" % (dt * 1e3), {k: round(v * 1e3, 1) for k, v in tm.items() if k.endswith("_s")})
print("raw", list(out))
pnames = ["mutation fast path", "mutation slow path", "structural: restage", "structural: decision, breaks",
          "structural: memo wipe", "structural: fast check", "structural: exact setup + walk", "structural: rounds",
          "other (trace, state)"]
tot = sum(out[24:33]) or 1
for i, nm in enumerate(pnames):
    print("  %-32s %5.1f%%   %.0f ticks per chain-step" % (nm, 100.0 * out[24 + i] / tot, out[24 + i] / ws))
print("total ticks per chain-step %.0f (s_memtime: 100 MHz)" % (tot / ws))
print("nested: cache probe %.0f ticks per chain-step; probe + evaluation %.0f; request-lanes evaluated %.2f per chain-step -> %.0f ticks each" % (
    out[33] / ws, out[34] / ws, out[35] / ws, (out[34] - out[33]) / max(out[35], 1)))
print("sub-timers (ticks per chain-step): mutation staging + permutation %.0f; context acquire %.0f; spec_eval inside mutation rounds %.0f; inside structural rounds %.0f" % (
    out[48] / ws, out[49] / ws, out[52] / ws, out[53] / ws))
if out[8]:  # (event counters: the `stats` build)
    print("mutation: wave-calls %d  on the slow path %.3f  rounds per slow-path call %.2f; requests %d misses %d (%.1f %%)" % (
        out[8], out[9] / out[8], out[11] / max(out[9], 1), out[0], out[1], 100.0 * out[1] / max(out[0], 1)))
    for nm, b, w in (("recombination", 12, 20), ("dosage (both kinds)", 16, 21)):
        n = max(out[b], 1)
        print("%s: wave-calls %d  executing %.3f  needing rounds %.3f  rounds per call with rounds %.2f" % (
            nm, out[b], out[b + 1] / n, out[b + 2] / n, out[b + 3] / max(out[w], 1)))
    slow = max(out[9], 1)
    print("per slow-path mutation call: %.0f ticks; cache probe %.0f + evaluations %.0f of them (mutation + structural together), %.2f request-lanes evaluated" % (
        out[25] / slow, out[33] / slow, (out[34] - out[33]) / slow, out[35] / slow))
