#!/bin/bash
# dictionary recursion on the card: forced-route sweeps, then S-32G (auto route) with stage log
tag=$1
for env in "PFP_DICT_REC=1" "PFP_DICT_REC=1 PFP_DICT_REC_P2=5 PFP_PARSE_REC_TILE_ROWS=20 PFP_PARSE_REC=1"; do
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_dr_random.log 2>&1 || { tail -5 gpurun_out/${tag}_dr_random.log; exit 1; }
  tail -1 gpurun_out/${tag}_dr_random.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_dr_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_dr_medium.log; exit 1; }
  tail -1 gpurun_out/${tag}_dr_medium.log
done
PFP_VERBOSE=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_verbose.json 2> gpurun_out/${tag}_bench_verbose.err || { tail -20 gpurun_out/${tag}_bench_verbose.err; exit 1; }
grep -n "recursive dictionary\|D2 sorted\|P2 sorted\|dictionary assembled\|through the global" gpurun_out/${tag}_bench_verbose.err | head -8
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'])"
