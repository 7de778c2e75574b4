"""Profiling helper (not part of the product): regions and events of the steady-state sampler (kernel 4, K = 4) from
the counters build (make -C mchap_amd/csrc phases).  python tools/lane_stats.py [loci] [steps]"""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCHAP_HIP_KERNEL", "4")
import torch
from mchap_amd import _lib
_lib.SO = os.environ.get("MCHAP_HIP_LIB") or os.path.join(_lib.CSRC, "libmchap_hip_phases.so")
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=steps, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
out = (C.c_ulonglong * 24)()
L = _lib.lib()
L.mchap_debug_lane_stats(out, 1)
t = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t
L.mchap_debug_lane_stats(out, 1)
lanes = int(os.environ.get("MCHAP_HIP_LANES", "0"))
print("loci", U, "steps", steps, "lanes", lanes, "time %.3f s" % dt, L.mchap_last_sampler_name().decode())
names = ["record + loop head", "mutation: Philox + test", "mutation: serving", "structural: fast loop", "(after kind loop)",
         "fill rounds", "(tail)", "structural: exact walk + serving"]
tot = sum(out[0:8]) or 1
for i, nm in enumerate(names):
    print("  %-34s %5.1f%%  %12d ticks" % (nm, 100.0 * out[i] / tot, out[i]))
cs = U * 2 * steps
print("per chain-step: mutation serves %.4f, fast-loop iterations/wave %d, refills %d, exact entries %d, required serves %.4f, fill serves %.4f" % (
    out[10] / cs, out[11], out[12], out[13], out[14] / cs, out[15] / cs))
print("settle kernel: chains parked %d (mean step %.1f), chains finished in it %d; steady kernel: chain runs %d, steps %d (%.1f per run), handed back %d" % (
    out[16], out[17] / max(out[16], 1), out[19], out[20], out[18], out[18] / max(out[20], 1), out[21]))
print("mutation serves: probe %.0f ticks, probe + evaluations %.0f ticks, %.1f evaluations per serve -> %.0f ticks per evaluation" % (
    out[20] / max(out[10], 1), out[21] / max(out[10], 1), out[22] / max(out[10], 1), (out[21] - out[20]) / max(out[22], 1)))
print("structural serves: probe %.0f ticks, probe + evaluations %.0f ticks, %.1f evaluations per serve -> %.0f ticks per evaluation" % (
    out[8] / max(out[14], 1), out[9] / max(out[14], 1), out[23] / max(out[14], 1), (out[9] - out[8]) / max(out[23], 1)))
