"""Measurement aid (not part of the product): the phased sampler (kernel 5) against the speculative sampler (kernel 3)
on the headline workload -- identical traces required, sampler time per setting of the pipeline's knobs.
    python tools/pipe_sweep.py [loci]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads, _, _ = synth_units(U)


def run(kernel, env=None, reps=3):
    for k in list(os.environ):
        if k.startswith("MCHAP_HIP_"):
            del os.environ[k]
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42, kernel=kernel)
    b = DenovoDeviceBatch(model, reads)
    b.time_sampler(True)
    ms = []
    for _ in range(reps):
        b.run()
        torch.cuda.synchronize()
        ms.append(b.sampler_ms())
    name = b.sampler_name
    out = (b.d_trace.cpu().numpy().copy(), b.d_llks.cpu().numpy().copy(), b.d_status.cpu().numpy().copy())
    del b
    return name, ms, out


name, ms, ref = run(3)
print("%-70s %s" % (name, " ".join("%.2f" % m for m in ms)), flush=True)
settings = [{}, {"MCHAP_HIP_PIPE_FIRST": 4}, {"MCHAP_HIP_PIPE_FIRST": 2}, {"MCHAP_HIP_ROUNDS": 0}, {"MCHAP_HIP_ROUNDS": 4}]
if len(sys.argv) > 2:
    # e.g. "MCHAP_HIP_PIPE_FIRST=4,MCHAP_HIP_COAST_LANES=8;MCHAP_HIP_ROUNDS=1"
    settings = [dict(kv.split("=") for kv in grp.split(",") if kv) for grp in sys.argv[2].split(";")]
for env in settings:
    name, ms, out = run(5, env)
    same = all(np.array_equal(a, b) for a, b in zip(ref, out))
    print("%-70s %s  %s  %s" % (name, " ".join("%.2f" % m for m in ms), env, "same traces" if same else "TRACES DIFFER"), flush=True)
    if not same:
        d = np.argwhere(ref[0] != out[0])
        print("   first differing trace word:", d[:3].tolist(), " llk diffs:", int((ref[1] != out[1]).sum()), " status diffs:", int((ref[2] != out[2]).sum()))
