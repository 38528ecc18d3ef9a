#!/bin/bash
# SQ counters of k_dedup_insert on S-3G (distinct phrases), cooperative against per-lane
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PFP_TEST_HOOKS=1
for v in 1 0; do
  export PFP_DEDUP_VARIANT=$v
  rm -rf /tmp/pmc_v$v
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_v$v -- python3 $root/tools/parse_bench.py --workload S-3G --reps 1 > $root/gpurun_out/${tag}_v$v.log 2>&1; echo "pmc v$v rc=$?"
  f=$(find /tmp/pmc_v$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && { head -1 "$f"; grep "dedup_insert" "$f" | grep -v "insert_long"; } > $root/gpurun_out/${tag}_v${v}_counters.csv; wc -l "$f"; sed -n 2,3p "$f" | cut -c1-300
  [ -n "$f" ] && python3 - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "k_dedup_insert" in k and "long" not in k:
        acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:40], r["Counter_Name"])] += 1
for k, d in acc.items():
    print("variant", sys.argv[2], k, {c: "%.3g" % v for c, v in sorted(d.items())})
PY
done
