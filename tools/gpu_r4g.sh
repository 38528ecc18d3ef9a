#!/bin/bash
# "follow the previous match" in the text de-duplication: forced-route sweeps on the card, then S-32G (auto route) with its hit rate
tag=$1
for env in "PFP_DEDUP_FOLLOW=50 PFP_DEDUP_FOLLOW_PROBE=100 PFP_DEDUP_FOLLOW_PCT=0" "PFP_DEDUP_FOLLOW=700 PFP_DEDUP_FOLLOW_PROBE=2000 PFP_DEDUP_FOLLOW_PCT=5 PFP_DICT_REC=1 PFP_PARSE_REC=1"; do
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_fw_random.log 2>&1 || { tail -5 gpurun_out/${tag}_fw_random.log; exit 1; }
  tail -1 gpurun_out/${tag}_fw_random.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_fw_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_fw_medium.log; exit 1; }
  tail -1 gpurun_out/${tag}_fw_medium.log
done
PFP_VERBOSE=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_verbose.json 2> gpurun_out/${tag}_bench_verbose.err || { tail -20 gpurun_out/${tag}_bench_verbose.err; exit 1; }
grep -n "de-duplication" gpurun_out/${tag}_bench_verbose.err | head -3
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'], d['roofline'])"
PFP_TEST_HOOKS=1 PFP_DEDUP_FOLLOW=0 timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g_nofollow.json 2> gpurun_out/${tag}_bench_s32g_nofollow.err || { tail -20 gpurun_out/${tag}_bench_s32g_nofollow.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g_nofollow.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['roofline'])"
timeout -k 10 600 python bench.py --workload S-3G --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s3g.json 2> gpurun_out/${tag}_bench_s3g.err || { tail -20 gpurun_out/${tag}_bench_s3g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s3g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'])"
