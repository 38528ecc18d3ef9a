#!/bin/bash
# column order for the parse-level phrase table (k_rs_dedup): parity, bench
tag=$1
for env in "PFP_PARSE_REC=1 PFP_DEDUP_PERIOD=2 PFP_DEDUP_CHUNK=1" "PFP_PARSE_REC=1 PFP_PARSE_REC_DEPTH=2 PFP_DEDUP_PERIOD=3 PFP_DEDUP_VARIANT=1 PFP_DICT_REC=1"; do
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_random.log 2>&1 || { tail -5 gpurun_out/${tag}_random.log; exit 1; }
  tail -1 gpurun_out/${tag}_random.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 5 6 --medium 10 --child > gpurun_out/${tag}_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_medium.log; exit 1; }
  tail -1 gpurun_out/${tag}_medium.log
done
timeout -k 10 900 python -m pytest tests/test_recsort.py tests/test_gpu_routes.py -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest.log
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'])"
PFP_TEST_HOOKS=1 PFP_DEDUP_PERIOD=-1 timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g_textorder.json 2> gpurun_out/${tag}_bench_s32g_textorder.err || { tail -20 gpurun_out/${tag}_bench_s32g_textorder.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g_textorder.json').read()); print('text order', round(d['ms_per_step'],1), d['stage_ms'])"
timeout -k 10 600 python tools/big_check_routes.py --workload S-32G > gpurun_out/${tag}_big_routes_S-32G.log 2>&1 || { tail -20 gpurun_out/${tag}_big_routes_S-32G.log; exit 1; }
tail -1 gpurun_out/${tag}_big_routes_S-32G.log | cut -c1-120
