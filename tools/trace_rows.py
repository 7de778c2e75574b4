"""Measurement aid: the launches of a rocprofv3 --kernel-trace CSV in time order (name, duration, grid)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    n = r["Kernel_Name"]
    if pat in n:
        print("%-64s %9.3f ms  grid %s" % (n[:64], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size"))))
