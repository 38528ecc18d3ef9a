#!/bin/bash
# text de-duplication, round 4: cooperative representative reads + column order -- parity sweeps, chunk sweep, bench, full-size equality with the old routes
tag=$1
for env in "PFP_DEDUP_PERIOD=3 PFP_DEDUP_CHUNK=2" "PFP_DEDUP_VARIANT=0 PFP_DEDUP_PERIOD=2" "PFP_DEDUP_PERIOD=-1"; do
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 900 1000 --count 25 --child > gpurun_out/${tag}_dd_random.log 2>&1 || { tail -5 gpurun_out/${tag}_dd_random.log; exit 1; }
  tail -1 gpurun_out/${tag}_dd_random.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/stress_random.py --seeds 3 4 --medium 10 --child > gpurun_out/${tag}_dd_medium.log 2>&1 || { tail -5 gpurun_out/${tag}_dd_medium.log; exit 1; }
  tail -1 gpurun_out/${tag}_dd_medium.log
done
timeout -k 10 900 python -m pytest tests/test_gpu_routes.py -m gpu -x -q > gpurun_out/${tag}_pytest_routes.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest_routes.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest_routes.log
for ch in 8 16 64 128; do
  echo "chunk $ch" >> gpurun_out/${tag}_chunk.log
  PFP_TEST_HOOKS=1 PFP_DEDUP_CHUNK=$ch timeout -k 10 300 python tools/parse_bench.py --reps 2 >> gpurun_out/${tag}_chunk.log 2>&1 || { tail -5 gpurun_out/${tag}_chunk.log; exit 1; }
done
grep "chunk\|^rep 2" gpurun_out/${tag}_chunk.log | cut -c1-70
timeout -k 10 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['full_size_order_check'], d['roofline'])"
for wl in S-32G S-3G; do
  timeout -k 10 600 python tools/big_check_routes.py --workload $wl > gpurun_out/${tag}_big_routes_$wl.log 2>&1 || { tail -20 gpurun_out/${tag}_big_routes_$wl.log; exit 1; }
  tail -1 gpurun_out/${tag}_big_routes_$wl.log | cut -c1-120
done
