#!/bin/bash
# S-3G (distinct phrases): the text de-duplication with the wave's new entries counted together; variants side by side
tag=$1
for v in 1 0; do
  PFP_TEST_HOOKS=1 PFP_DEDUP_VARIANT=$v timeout -k 10 400 python tools/parse_bench.py --workload S-3G --reps 2 > gpurun_out/${tag}_s3g_v$v.log 2>&1 || { tail -5 gpurun_out/${tag}_s3g_v$v.log; exit 1; }
  grep "^rep 2" gpurun_out/${tag}_s3g_v$v.log | cut -c1-200
done
PFP_TEST_HOOKS=1 timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_random.log 2>&1 || { tail -5 gpurun_out/${tag}_random.log; exit 1; }
tail -1 gpurun_out/${tag}_random.log
PFP_TEST_HOOKS=1 PFP_DEDUP_PERIOD=3 PFP_DEDUP_TABLE_LOG2=8 timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_medium.log; exit 1; }
tail -1 gpurun_out/${tag}_medium.log
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'])"
timeout -k 10 600 python bench.py --workload S-3G --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s3g.json 2> gpurun_out/${tag}_bench_s3g.err || { tail -20 gpurun_out/${tag}_bench_s3g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s3g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'])"
