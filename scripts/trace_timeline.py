#!/usr/bin/env python3
"""Print the last N kernel dispatches of a rocprofv3 kernel-trace CSV as a timeline (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:28s} q{r.get('Queue_Id','?'):>3s} start {s:9.1f} end {e:9.1f} dur {e-s:8.1f} grid {r.get('Grid_Size','?')}")
