#!/bin/bash
tag=$1
PFP_VERBOSE=1 PFP_TEST_HOOKS=1 PFP_DEDUP_FOLLOW=5100 PFP_DEDUP_FOLLOW_PROBE=10000 timeout -k 10 200 python tools/follow_check.py 4 500000 2>&1 | grep -E "de-dup|n="
PFP_VERBOSE=1 timeout -k 10 200 python tools/follow_check.py 20 32000000 u64 2>&1 | grep -E "de-dup|n="
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'], d['roofline'])"
PFP_TEST_HOOKS=1 PFP_DEDUP_FOLLOW=0 timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g_nofollow.json 2> gpurun_out/${tag}_bench_s32g_nofollow.err || { tail -20 gpurun_out/${tag}_bench_s32g_nofollow.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g_nofollow.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['roofline'])"
for env in "PFP_DEDUP_FOLLOW=50 PFP_DEDUP_FOLLOW_PROBE=100 PFP_DEDUP_FOLLOW_PCT=0" "PFP_DEDUP_FOLLOW=700 PFP_DEDUP_FOLLOW_PROBE=2000 PFP_DEDUP_FOLLOW_PCT=5 PFP_DICT_REC=1 PFP_PARSE_REC=1"; do
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_fw_random.log 2>&1 || { tail -5 gpurun_out/${tag}_fw_random.log; exit 1; }
  tail -1 gpurun_out/${tag}_fw_random.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_fw_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_fw_medium.log; exit 1; }
  tail -1 gpurun_out/${tag}_fw_medium.log
done
