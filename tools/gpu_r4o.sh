#!/bin/bash
# text de-duplication on collections with more distinct phrases: cooperative (1) against per-lane (0), column order on (default) / off (1xxx)
tag=$1
for wl in S-20x32M S-100x32M; do
  PFP_TEST_HOOKS=1 timeout -k 10 400 python tools/parse_bench.py --workload $wl --variants 1 0 1001 1000 > gpurun_out/${tag}_$wl.log 2>&1 || { tail -5 gpurun_out/${tag}_$wl.log; exit 1; }
  echo $wl; grep "^rep" gpurun_out/${tag}_$wl.log | cut -c1-130
done
