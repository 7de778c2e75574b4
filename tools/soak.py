"""Soak aid: the differential fuzz of tests/fuzz_kernels.py (and tests/fuzz_call.py: the `mchap call` sampler) over seeds and corners the test suite does not hold
(longer runs, deeper reads at high ploidy, more positions).  Needs a GPU.
    python tools/soak.py [first_seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import fuzz_kernels
import fuzz_call

s = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bad = 0
bad += fuzz_kernels.run(40, s)
bad += fuzz_kernels.run(10, s + 1, ploidies=(5, 6, 7, 8), read_depths=(260, 400, 700, 1000), tempering=False, max_pos=16)
bad += fuzz_kernels.run(10, s + 2, ploidies=(2, 3, 4), read_depths=(20, 100, 300, 600), tempering=False, max_pos=20)
bad += fuzz_call.run(150, s + 3)  # the `mchap call` sampler: wavefront kernel + a lane per settled chain
print("SOAK FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
