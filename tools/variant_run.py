import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
envs = [e for e in sys.argv[1:]] or [""]
for so in sorted(glob.glob(os.path.join(root, "mchap_amd/csrc/dbg/*.so"))):
    for e in envs:
        print(os.path.basename(so), e, flush=True)
        env = dict(os.environ, MCHAP_HIP_LIB=so)
        for kv in e.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        subprocess.call([sys.executable, os.path.join(root, "tools/debug_one.py")], env=env)
