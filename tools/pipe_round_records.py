"""Measurement / debugging aid: the phased sampler stopped after each round (MCHAP_HIP_PIPE_STOP): hand-over records, interval tables and
trace rows of one docs/example unit against kernel 3.   python tools/pipe_round_records.py locus002 progeny015 [flags]"""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, application, io, _lib
from mchap_amd.assemble import unpack_trace
from mchap_amd.device import DenovoRaggedBatch
samples, targets, variants, matrices, contigs = application.load_matrices("tests/golden/example_biparental.npz")
source = application.MatrixSource(samples, matrices)
name, sample = sys.argv[1], sys.argv[2]
t = [x for x in targets if x[3] == name][0]
locus = io.DenovoLocus(t[0], t[1], t[2], t[3], variants, "N" * (t[2] - t[1]), sequence_known=False)
sr = source.reads(locus, sample)
M = len(locus.positions)
STEPS = 600
os.environ["MCHAP_HIP_KERNEL"] = "3"
ref = DenovoMCMC(ploidy=4, n_alleles=list(locus.n_alleles), steps=STEPS, chains=2, random_seed=42).fit_batch([sr["dists"]], [sr["counts"]], stream_ids=[0])[0]
os.environ["MCHAP_HIP_KERNEL"] = "5"
os.environ["MCHAP_HIP_TEST_KERNELS"] = "1"
os.environ["MCHAP_HIP_FLAGS"] = sys.argv[3] if len(sys.argv) > 3 else "64"
os.environ["MCHAP_HIP_ROUNDS"] = "6"
E = M * (M + 1) // 2
for stop in (3, 5, 7):
    os.environ["MCHAP_HIP_PIPE_STOP"] = str(stop)
    model = DenovoMCMC(ploidy=4, n_alleles=list(locus.n_alleles), steps=STEPS, chains=2, random_seed=42)
    b = DenovoRaggedBatch(model, [dict(reads=sr["dists"], counts=sr["counts"], n_alleles=locus.n_alleles, ploidy=4, inbreeding=None, stream_id=0)])
    b.run(300)
    torch.cuda.synchronize()
    L = _lib.lib()
    rec = np.zeros((2, 16), dtype=np.uint64)
    cnt = np.zeros(8, dtype=np.int32)
    memo = np.zeros((2, 2, E))
    f = L.mchap_debug_pipe_records; f.restype = C.c_int
    assert f(C.byref(b.cfg), 1, _lib.ptr(b.units_host), C.c_void_p(b.d_ws.data_ptr()), _lib.ptr(rec), _lib.ptr(cnt)) == 0
    f2 = L.mchap_debug_pipe_memo; f2.restype = C.c_int
    assert f2(C.byref(b.cfg), 1, _lib.ptr(b.units_host), C.c_void_p(b.d_ws.data_ptr()), _lib.ptr(memo)) == 0
    fixed = b.d_fixed.cpu().numpy()
    lk = b.d_llks.cpu().numpy().reshape(2, STEPS)
    for c in range(2):
        upto = int(rec[c, 12] & np.uint64(0xFFFFFFFF))
        bad = np.flatnonzero(lk[c, :upto] != ref.llks[c, :upto])
        print("   stop", stop, "chain", c, "rows < record.step that differ from ref:", bad.tolist()[:50])
    for c in range(2):
        step = int(rec[c, 12] & np.uint64(0xFFFFFFFF)); mvalid = int(rec[c, 12] >> np.uint64(32))
        llk = rec[c, 8:9].view(np.float64)[0]
        g = unpack_trace(rec[c, :4][None], fixed, 2)[0]
        ok_g = step >= 1 and step <= STEPS and sorted(map(bytes, g.astype(np.int8))) == sorted(map(bytes, ref.genotypes[c, min(step, STEPS) - 1].astype(np.int8)))
        print("stop", stop, "chain", c, "record: step", step, "mvalid", mvalid, "llk %.6f" % llk, "ref llk at step-1: %.6f" % ref.llks[c, min(step, STEPS) - 1], "genotype ok" if ok_g else "GENOTYPE DIFFERS",
              "mlo %.3g 1-mhi %.3g" % (rec[c, 10:11].view(np.float64)[0], 1 - rec[c, 11:12].view(np.float64)[0]), "memo nan %d neg %d max %.3g" % (np.isnan(memo[c]).sum(), (memo[c] < 0).sum(), np.nanmax(memo[c])), "counts", cnt.tolist(), flush=True)
