#!/bin/bash
# does the size of the phrase table (pages a random read may touch) set the time of k_dedup_insert?  (follow path off)
tag=$1
for lg in 21 22 23 24 25 27; do
  echo "table 2^$lg" >> gpurun_out/${tag}_table.log
  PFP_TEST_HOOKS=1 PFP_DEDUP_FOLLOW=0 PFP_DEDUP_TABLE_LOG2=$lg timeout -k 10 300 python tools/parse_bench.py --reps 2 >> gpurun_out/${tag}_table.log 2>&1 || { tail -5 gpurun_out/${tag}_table.log; exit 1; }
done
grep "table\|^rep 2" gpurun_out/${tag}_table.log | cut -c1-110
