"""Measurement aid: `mchap call` sampler (call_mcmc_kernel) time against the number of steps and known haplotypes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mchap_amd import _lib
from mchap_amd.synth import synth_units

def run(U, K, H, M, R, S, Cn, step_type=0):
    rng = np.random.default_rng(9)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=700)
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(8 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool)
        tr = np.unique(truth[u], axis=0)
        rest = [p_ for p_ in pool if not any(np.array_equal(p_, t) for t in tr)][: H - len(tr)]
        hs = np.concatenate([tr, np.array(rest, np.int8)]); rng.shuffle(hs); haps[u] = hs
    dev = torch.device("cuda")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    d_reads, d_haps, d_F, d_sid = d(reads), d(haps), d(np.full(U, 0.1)), d(np.arange(U, dtype=np.int64))
    d_g = torch.empty(U * Cn * S * K, dtype=torch.int64, device=dev); d_l = torch.empty(U * Cn * S, dtype=torch.float64, device=dev)
    d_st = torch.empty(U, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = int(L.mchap_call_mcmc_workspace_bytes_for(U, R, H, K, S, Cn)); d_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def once():
        _lib.check(L.mchap_call_mcmc_batch_device(U, p(d_reads), R, M, 2, None, p(d_haps), H, K, 1, p(d_F), None, None, p(d_sid), S, Cn, step_type,
                                                  C.c_uint64(42), p(d_g), p(d_l), p(d_st), p(d_ws), C.c_int64(ws), C.c_void_p(st)))
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); once(); e1.record(); torch.cuda.synchronize()
    g = d_g.cpu().numpy().reshape(U, Cn, S, K)
    moves = (np.diff(g, axis=2) != 0).any(axis=3).mean()
    print("U %d K %d H %2d R %d steps %4d type %d: %.1f ms  (ws %.0f MB; steps that change the genotype %.3f)" % (U, K, H, R, S, step_type, e0.elapsed_time(e1), ws / 1e6, moves), flush=True)

for S in (100, 500, 2000):
    run(4096, 4, 16, 8, 200, S, 2)
run(4096, 4, 4, 8, 200, 2000, 2)
run(4096, 4, 16, 8, 50, 2000, 2)
run(1024, 4, 16, 8, 200, 2000, 2)
run(4096, 4, 16, 8, 200, 2000, 2, 1)
