#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> compact timeline: start_us,dur_us,gap_us,kernel (short name), in dispatch order"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
prev_end = t0
with open(sys.argv[2], "w") as f:
    f.write("start_us,dur_us,gap_us,kernel\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("pfp::", "").split("(")[0].replace("void ", "")[:48]
        f.write("%.1f,%.1f,%.1f,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
        prev_end = max(prev_end, e)
