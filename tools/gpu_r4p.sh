#!/bin/bash
# end of round 4: S-50G on the final code, its full-size equality with the old routes, the 8-rank shard replay on one card
tag=$1
timeout -k 10 600 python bench.py --workload S-50G --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s50g.json 2> gpurun_out/${tag}_bench_s50g.err || { tail -20 gpurun_out/${tag}_bench_s50g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s50g.json').read()); print(round(d['ms_per_step'],1), round(d['value'],1), d['stage_ms'], d['full_size_order_check'])"
timeout -k 10 600 python tools/big_check_routes.py --workload S-50G > gpurun_out/${tag}_big_routes_S-50G.log 2>&1 || { tail -20 gpurun_out/${tag}_big_routes_S-50G.log; exit 1; }
tail -1 gpurun_out/${tag}_big_routes_S-50G.log | cut -c1-120
timeout -k 10 900 python tools/shard_sim.py > gpurun_out/${tag}_shard_sim_8ranks_s32g.log 2>&1 || { tail -20 gpurun_out/${tag}_shard_sim_8ranks_s32g.log; exit 1; }
tail -6 gpurun_out/${tag}_shard_sim_8ranks_s32g.log | cut -c1-220
