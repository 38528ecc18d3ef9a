#!/bin/bash
# dictionary recursion: route tests on the card, full-size equality with the old routes, kernel trace of the bench
tag=$1
timeout -k 10 900 python -m pytest tests/test_recsort.py tests/test_gpu_routes.py -m gpu -x -q > gpurun_out/${tag}_pytest_routes.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest_routes.log; exit 1; }
tail -2 gpurun_out/${tag}_pytest_routes.log
for wl in S-32G S-50G S-3G; do
  timeout -k 10 600 python tools/big_check_routes.py --workload $wl > gpurun_out/${tag}_big_routes_$wl.log 2>&1 || { tail -20 gpurun_out/${tag}_big_routes_$wl.log; exit 1; }
  tail -1 gpurun_out/${tag}_big_routes_$wl.log | cut -c1-200
done
tools/gpu_trace.sh ${tag}_trace --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end || { tail -20 gpurun_out/${tag}_trace.err; exit 1; }
head -40 gpurun_out/${tag}_trace_kernel_stats.csv | cut -c1-160
