"""Bench every debug build under mchap_amd/csrc/dbg416/ (development aid)."""
import glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:]
for so in sorted(glob.glob(os.path.join(root, "mchap_amd/csrc/dbg416/*.so"))):
    env = dict(os.environ, MCHAP_HIP_LIB=so)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + extra,
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
        print("%-14s %9.0f loci/s  kernel %.2f ms" % (os.path.basename(so), d["value"], d["roofline"]["kernel_ms"]), flush=True)
    except Exception as e:
        print(os.path.basename(so), "failed", e, out[-3:])
