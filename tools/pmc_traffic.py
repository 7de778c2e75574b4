"""Turn the rocprofv3 --pmc CSVs of tools/profile_round.sh into profiles/<tag>_pmc_traffic.json.

FETCH_SIZE / WRITE_SIZE are in KiB.  /opt/skills/guides/MI355X_MICROARCH.md ("HBM") calibrates them for 16-byte-
per-lane streams only (FETCH_SIZE reports half of the bytes there) and asks for a calibration in one's own access
pattern otherwise: tools/pmc_calib.hip streams 1 GiB with the sampler's width (8 bytes per lane) in the same
profiling session, and the measured bytes-per-counter-unit factors are applied to the sampler's counters.

    python tools/pmc_traffic.py <tag> <loci> <mcmc_steps> <chains>
"""
import collections
import csv
import glob
import json
import sys

tag, loci, steps, chains = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
SAMPLERS = ("denovo_spec", "denovo_coast", "denovo_fillw", "denovo_simt_kernel", "denovo_mcmc")


def per_launch(dirname, match):
    """Counter totals per sampler call: a call of the phased sampler is several dispatches (speculative kernel phases +
    coasting kernel), so the matching dispatches are summed and divided by the number of calls -- one
    denovo_prepare_kernel dispatch precedes each; the calibration kernels are one dispatch per call."""
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, dirname)):
        seen = set()
        n_prepare = set()
        names = set()
        for row in csv.DictReader(open(f)):
            if "denovo_prepare_kernel" in row["Kernel_Name"]:
                n_prepare.add(row["Dispatch_Id"])
            if any(m in row["Kernel_Name"] for m in match):
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                seen.add(row["Dispatch_Id"])
                names.add(row["Counter_Name"])
        for name in names:
            calls[name] += len(n_prepare) if n_prepare else len(seen)
    return {k: tot[k] / max(calls[k], 1) for k in tot}, dict(calls)


fetch, nf = per_launch("pmc_fetch", SAMPLERS)
write, nw = per_launch("pmc_write", SAMPLERS)
cal_bytes = float((1 << 27) * 8)
cf, _ = per_launch("cal_fetch", ("calib_read_f64",))
cw, _ = per_launch("cal_write", ("calib_write_f64",))
f_factor = cal_bytes / (cf["FETCH_SIZE"] * 1024.0) if cf.get("FETCH_SIZE") else None
w_factor = cal_bytes / (cw["WRITE_SIZE"] * 1024.0) if cw.get("WRITE_SIZE") else None
fs, ws = fetch.get("FETCH_SIZE", 0.0), write.get("WRITE_SIZE", 0.0)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16  # noqa: E402  (the summary names the sources it was taken from: bench.py quotes no other)

out = {
    "loci": loci, "mcmc_steps": steps, "chains": chains, "kernel_sources_sha16": kernel_sources_sha16(),
    "FETCH_SIZE_KiB_per_launch": fs, "WRITE_SIZE_KiB_per_launch": ws,
    "calibration": {
        "pattern": "8 bytes per lane, 512 contiguous bytes per wavefront, 1 GiB streamed (tools/pmc_calib.hip)",
        "bytes_per_FETCH_SIZE_byte": f_factor, "bytes_per_WRITE_SIZE_byte": w_factor,
    },
    "hbm_bytes_per_launch": ((f_factor or 2.0) * fs + (w_factor or 1.0) * ws) * 1024.0,
    "hbm_bytes_per_launch_guide_16B_rule": (2.0 * fs + ws) * 1024.0,
    "launches_averaged": [nf.get("FETCH_SIZE", 0), nw.get("WRITE_SIZE", 0)],
}
sq = {}
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    v, _ = per_launch(d, SAMPLERS)
    sq.update(v)
out["sq_counters_per_launch"] = sq
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
