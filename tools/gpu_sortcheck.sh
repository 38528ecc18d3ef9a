#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_sortcheck.sh <tag>
# quick loop for work on the refinement rounds: sort-route parity tests, then the profiled S-32G bench
tag=$1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort_routes or golden" > gpurun_out/${tag}_parity.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_parity.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_prof.sh ${tag}_prof_s32g --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end || exit 1
cut -c1-300 gpurun_out/${tag}_prof_s32g.json
python3 tools/kstats.py gpurun_out/${tag}_prof_s32g_kernel_stats.csv | grep -E "total|k_round" 
