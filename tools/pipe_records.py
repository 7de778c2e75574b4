"""Measurement aid (not part of the product): which chains the coasting kernel hands back, and when.  Runs the phased
sampler up to the end of its first coasting launch (MCHAP_HIP_PIPE_STOP) and reads the hand-over records.
    python tools/pipe_records.py [loci]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
os.environ["MCHAP_HIP_PIPE_STOP"] = "1"
os.environ["MCHAP_HIP_TEST_KERNELS"] = "1"  # mchap_debug_pipe_records lives in the parity suite's library
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42, kernel=5)
b = DenovoDeviceBatch(model, reads)
b.run()
torch.cuda.synchronize()
L = _lib.lib()
rec = np.zeros((U * 2, 16), dtype=np.uint64)
cnt = np.zeros(8, dtype=np.int32)
f = L.mchap_debug_pipe_records
f.restype = C.c_int
rc = f(C.byref(b.cfg), U, _lib.ptr(b.units_host), C.c_void_p(b.d_ws.data_ptr()), _lib.ptr(rec), _lib.ptr(cnt))
assert rc == 0, rc
step = (rec[:, 12] & np.uint64(0xFFFFFFFF)).astype(np.int64)
mvalid = (rec[:, 12] >> np.uint64(32)).astype(np.int64)
mlo = rec[:, 10].view(np.float64)
mhi = rec[:, 11].view(np.float64)
back = step < 1000
print("counts", cnt.tolist(), " handed back:", int(back.sum()), "of", len(step))
print("mvalid among handed back:", np.bincount(mvalid[back], minlength=2).tolist())
print("hand-back step percentiles:", np.percentile(step[back], [0, 10, 25, 50, 75, 90, 100]).tolist() if back.any() else None)
hb = np.flatnonzero(back)
for q in hb[:25]:
    print("  chain %6d step %4d mvalid %d mlo %.3e 1-mhi %.3e" % (q, step[q], mvalid[q], mlo[q], 1.0 - mhi[q]))
print("mlo > 1e-6 overall:", int((mlo > 1e-6).sum()), " 1-mhi > 1e-6:", int(((1 - mhi) > 1e-6).sum()))
