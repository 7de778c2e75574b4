"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_sq.py <dir> [<dir> ...] -> JSON {kernel: {counter: sum, "launches": n}}"""
import csv, glob, json, os, sys
out = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0].replace("void mchap::", "")
            e = out.setdefault(k, {})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            ids = e.setdefault("_ids", set())
            ids.add((path, row.get("Dispatch_Id")))
for k, e in out.items():
    e["launches_seen"] = len(e.pop("_ids"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16  # noqa: E402  (bench.py quotes a summary only for the sources it was taken from)
out["_meta"] = {"kernel_sources_sha16": kernel_sources_sha16()}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
