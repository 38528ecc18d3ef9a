#!/bin/bash
tag=$1
PFP_TEST_HOOKS=1 PFP_VERBOSE=1 timeout -k 10 600 python tools/parse_bench.py --variants 0 3 13 1003 403 0 3 > gpurun_out/${tag}_coop.log 2>&1 || { tail -5 gpurun_out/${tag}_coop.log; exit 1; }
grep "^rep\|k_dedup" gpurun_out/${tag}_coop.log | cut -c1-230
env PFP_TEST_HOOKS=1 PFP_DEDUP_VARIANT=3 timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_v3_random.log 2>&1 || { tail -5 gpurun_out/${tag}_v3_random.log; exit 1; }
tail -1 gpurun_out/${tag}_v3_random.log
env PFP_TEST_HOOKS=1 PFP_DEDUP_VARIANT=3 PFP_DEDUP_PERIOD=3 timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_v3_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_v3_medium.log; exit 1; }
tail -1 gpurun_out/${tag}_v3_medium.log
