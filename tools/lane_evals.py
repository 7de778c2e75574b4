"""Profiling helper: likelihood requests / evaluations per chain under the steady-state pipeline (counters build)."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCHAP_HIP_KERNEL", "4")
import torch
from mchap_amd import _lib
_lib.SO = os.path.join(_lib.CSRC, "libmchap_hip_stats.so")
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
out = (C.c_ulonglong * 24)()
L = _lib.lib()
L.mchap_debug_lane_stats(out, 1)
b.run(); torch.cuda.synchronize()
L.mchap_debug_lane_stats(out, 1)
print("kernel", L.mchap_last_sampler_name().decode())
print("per chain: requests %.1f, evaluations %.1f" % (out[0] / (2 * U), out[1] / (2 * U)))
