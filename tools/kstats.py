#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms per bench step: tools/kstats.py file.csv [steps_total]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step: %.1f" % (tot / 1e6 / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    name = r["Name"].replace("pfp::", "").replace("unsigned long", "u64").replace("unsigned int", "u32").replace("unsigned char", "u8")
    name = name.split("(")[0][:60]
    print("%-60s calls/step %7.1f  ms/step %8.2f  avg_us %9.1f  %5.1f%%" % (name, float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
