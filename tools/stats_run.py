"""Profiling helper (not part of the product): run the sampler from the counters build and print event counts."""
import ctypes as C
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import _lib
_lib.SO = os.environ.get("MCHAP_HIP_LIB") or os.path.join(_lib.CSRC, "libmchap_hip_stats.so")
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 200  # e.g. 12 with low qualities: chains that keep moving
K = int(os.environ.get("STATS_K", "4")); M = int(os.environ.get("STATS_M", "8")); CH = int(os.environ.get("STATS_CHAINS", "2"))
reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=n_reads, window=(8, M) if M > 8 else (4, 8), qual=(3, 20) if n_reads < 100 else (20, 40))
model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=steps, chains=CH, random_seed=42)
b = DenovoDeviceBatch(model, reads)
out = (C.c_ulonglong * 64)()
L = _lib.lib()
L.mchap_debug_stats(out, 1)
t = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t
L.mchap_debug_stats(out, 1)
print("units", U, "steps", steps, "time %.3f s" % dt)
print("raw", list(out))
print("requests %d  misses %d (%.2f%%)  probe-slots %d  requests/probe-slot %.3f" % (out[0], out[1], 100.0 * out[1] / max(out[0], 1), out[2], out[0] / max(out[2], 1)))
print("per chain-step: requests %.2f misses %.3f" % (out[0] / (U * CH * steps), out[1] / (U * CH * steps)))
names = ["mutation fast path", "mutation slow path", "structural: restage", "structural: decision, breaks",
         "structural: memo wipe", "structural: fast check", "structural: exact setup + walk", "structural: rounds",
         "other (trace, state)"]
tot = sum(out[24:33]) or 1
G = int(os.environ.get("MCHAP_HIP_GROUP", "64"))
waves = (U * CH + (64 // G) - 1) // (64 // G)
for i, nm in enumerate(names):
    print("  %-32s %5.1f%%   %.0f ticks per wave-step" % (nm, 100.0 * out[24 + i] / tot, out[24 + i] / max(waves * steps, 1)))
ws = max(waves * steps, 1)
print("nested (steps): cache probe %.0f ticks per wave-step; probe + co-operative evaluation %.0f ticks per wave-step; evaluations %.3f per wave-step -> %.0f ticks each" % (
    out[33] / ws, out[34] / ws, out[35] / ws, (out[34] - out[33]) / max(out[35], 1)))
print("sub-timers (ticks per wave-step): mutation staging + permutation %.0f; context acquire %.0f; spec_eval inside mutation rounds %.0f; inside structural rounds %.0f" % (
    out[48] / ws, out[49] / ws, out[52] / ws, out[53] / ws))
print("TOTAL ticks (100 MHz) per chain: steps %.0f  table completion %.0f" % (sum(out[24:33]) / (U * CH), sum(out[36:45]) / (U * CH)))
print("table completion: rounds %.0f ticks per chain; cache probe %.0f; probe + evaluation %.0f; request-lanes evaluated %.1f per chain -> %.0f ticks each" % (
    out[36 + 7] / (U * CH), out[45] / (U * CH), out[46] / (U * CH), out[47] / (U * CH), (out[46] - out[45]) / max(out[47], 1)))
print("steps: request-lanes evaluated %.1f per chain" % (out[35] / (U * CH)))
print("mutation: wave-calls %d  slow-path wave-calls %.3f  groups on slow path per wave-call %.3f  rounds per wave-call %.3f" % (
    out[8], out[9] / max(out[8], 1), out[10] / max(out[8], 1), out[11] / max(out[8], 1)))
print("decision contexts: acquires %d (new %d = %.1f%%)  rounds with contexts %d  sub-steps decided from a context %d, not in it %d" % (
    out[3], out[4], 100.0 * out[4] / max(out[3], 1), out[7], out[5], out[6]))
for nm, b, w in (("recombination", 12, 20), ("dosage (both kinds)", 16, 21)):
    n = max(out[b], 1)
    print("%s: wave-calls %d  groups executing per wave-call %.3f  groups needing rounds %.3f  wave-calls with rounds %.3f  rounds per wave-call %.3f" % (
        nm, out[b], out[b + 1] / n, out[b + 2] / n, out[w] / n, out[b + 3] / n))

f = (C.c_ulonglong * 64)()
if hasattr(L, "mchap_debug_fill_stats") and L.mchap_debug_fill_stats(f, 1) == 0 and f[11]:
    n = f[11]
    print("fill kernel per chain (%d chains): ticks list %.0f dedup %.0f stage %.0f eval %.0f totals %.0f | chunks %.2f uniques %.1f slots %.1f stagings %.2f batch-tiles %.2f" % (
        n, f[0] / n, f[1] / n, f[2] / n, f[3] / n, f[4] / n, f[6] / n, f[7] / n, f[8] / n, f[9] / n, f[10] / n))
