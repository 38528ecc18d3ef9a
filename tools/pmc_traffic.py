#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command into HBM bytes per launch per kernel.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic_latest.json S-32G

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters
are in KiB-like units of 1024 B as printed by rocprofv3; FETCH_SIZE counts 128-B streaming requests at 64 B, so it is
doubled.  The doubling over-counts kernels whose reads are dominated by isolated 64-B sector gathers (k_emit_slots,
k_emit): for those the raw value is the lower bound and the corrected value the upper bound; both are recorded."""
import collections, csv, glob, json, os, sys


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_counter_collection.csv under " + d)
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "S-32G"      # bench.py only quotes the file for the workload it was taken on
    ft, fn = load(fetch_dir, "FETCH_SIZE")
    wt, wn = load(write_dir, "WRITE_SIZE")
    res = {"_workload": workload}
    for k in sorted(set(ft) | set(wt), key=lambda k: -(2 * ft.get(k, 0) + wt.get(k, 0))):
        launches = max(fn.get(k, 0), wn.get(k, 0))
        if not launches:
            continue
        f_kb = ft.get(k, 0.0) / max(fn.get(k, 1), 1); w_kb = wt.get(k, 0.0) / max(wn.get(k, 1), 1)
        res[k] = {"launches": launches, "fetch_kb_raw_per_launch": f_kb, "write_kb_per_launch": w_kb,
                  "hbm_bytes_per_launch_raw": 1024.0 * (f_kb + w_kb),
                  "hbm_bytes_per_launch_corrected": 1024.0 * (2.0 * f_kb + w_kb)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in [kv for kv in res.items() if kv[0] != "_workload"][:12]:
        print("%-60s launches %5d  HBM bytes/launch raw %.3e corrected %.3e" % (k[:60], v["launches"], v["hbm_bytes_per_launch_raw"], v["hbm_bytes_per_launch_corrected"]))


if __name__ == "__main__":
    main()
