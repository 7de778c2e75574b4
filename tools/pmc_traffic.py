"""Turn the rocprofv3 --pmc CSVs of tools/profile_round.sh into profiles/<tag>_pmc_traffic.json.

HBM bytes per sampler launch = 2 x FETCH_SIZE (gfx950 reports half the bytes of a read stream,
/opt/skills/guides/MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KiB-units of the counter (x 1024)."""
import collections
import csv
import glob
import json
import sys

tag, loci, steps, chains = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
vals = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, d)):
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "denovo_spec" in row["Kernel_Name"] or "denovo_simt_kernel" in row["Kernel_Name"] or "denovo_mcmc" in row["Kernel_Name"]:
                per_dispatch[(row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        for (name, _), v in per_dispatch.items():
            vals[name].append(v)
fetch = sum(vals["FETCH_SIZE"]) / max(len(vals["FETCH_SIZE"]), 1)
write = sum(vals["WRITE_SIZE"]) / max(len(vals["WRITE_SIZE"]), 1)
out = {
    "loci": loci, "mcmc_steps": steps, "chains": chains,
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "correction": "FETCH_SIZE doubled (gfx950 reports half of a coalesced read stream); WRITE_SIZE as is",
    "launches_averaged": [len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])],
}
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(out)
