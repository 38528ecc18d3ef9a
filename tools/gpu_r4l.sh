#!/bin/bash
tag=$1
for ch in 40 53 80 159; do
  echo "chunk $ch" >> gpurun_out/${tag}_chunk.log
  PFP_TEST_HOOKS=1 PFP_DEDUP_CHUNK=$ch timeout -k 10 300 python tools/parse_bench.py --reps 2 >> gpurun_out/${tag}_chunk.log 2>&1 || { tail -5 gpurun_out/${tag}_chunk.log; exit 1; }
done
grep "chunk\|^rep 2" gpurun_out/${tag}_chunk.log | cut -c1-70
