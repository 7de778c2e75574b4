"""Soak aid: tests/fuzz_moving.py (shallow low-quality units whose chains never settle, random shapes, the phased and the
speculative sampler against the oracle) over new seeds.  Needs a GPU.
    python tools/soak_moving.py [cases] [first_seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import fuzz_moving

bad = fuzz_moving.run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 9000)
sys.exit(1 if bad else 0)
