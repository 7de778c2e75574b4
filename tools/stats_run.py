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
reads, _, _ = synth_units(U, n_reads=n_reads, qual=(3, 20) if n_reads < 100 else (20, 40))
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=steps, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
out = (C.c_ulonglong * 24)()
L = _lib.lib()
L.mchap_debug_stats(out, 1)
t = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t
L.mchap_debug_stats(out, 1)
print("units", U, "steps", steps, "time %.3f s" % dt)
print("requests %d  misses %d (%.2f%%)  probe-slots %d  requests/probe-slot %.3f" % (out[0], out[1], 100.0 * out[1] / max(out[0], 1), out[2], out[0] / max(out[2], 1)))
print("per chain-step: requests %.2f misses %.3f" % (out[0] / (U * 2 * steps), out[1] / (U * 2 * steps)))
names = ["mutation fast path", "mutation slow path", "structural: restage", "structural: decision, breaks",
         "structural: memo wipe", "structural: fast check", "structural: exact setup + walk", "structural: rounds",
         "other (trace, state)"]
tot = sum(out[3:12]) or 1
G = int(os.environ.get("MCHAP_HIP_GROUP", "16"))
waves = (U * 2 + (64 // G) - 1) // (64 // G)
for i, nm in enumerate(names):
    print("  %-32s %5.1f%%   %.0f ticks per wave-step" % (nm, 100.0 * out[3 + i] / tot, out[3 + i] / max(waves * steps, 1)))
ws = max(waves * steps, 1)
print("nested: cache probe %.0f ticks per wave-step; probe + co-operative evaluation %.0f ticks per wave-step; evaluations %.3f per wave-step -> %.0f ticks each" % (
    out[12] / ws, out[13] / ws, out[14] / ws, (out[13] - out[12]) / max(out[14], 1)))
print("mutation: wave-calls %d  slow-path wave-calls %.3f  groups on slow path per wave-call %.3f  rounds per wave-call %.3f" % (
    out[8], out[9] / max(out[8], 1), out[10] / max(out[8], 1), out[11] / max(out[8], 1)))
for nm, b, w in (("recombination", 12, 20), ("dosage (both kinds)", 16, 21)):
    n = max(out[b], 1)
    print("%s: wave-calls %d  groups executing per wave-call %.3f  groups needing rounds %.3f  wave-calls with rounds %.3f  rounds per wave-call %.3f" % (
        nm, out[b], out[b + 1] / n, out[b + 2] / n, out[w] / n, out[b + 3] / n))
