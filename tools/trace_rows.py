"""Measurement aid: the launches of a rocprofv3 --kernel-trace CSV in time order (name, duration, grid)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if pat in n:
        if t0 is None:
            t0 = int(r["Start_Timestamp"])
        print("%-64s %9.3f ms  grid %-8s start %9.3f ms  lds %s scratch %s vgpr %s" % (
            n[:64], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size")),
            (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("LDS_Block_Size", "?"), r.get("Scratch_Size", "?"), r.get("VGPR_Count", "?")))
