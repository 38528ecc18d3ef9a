#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_quick.sh <tag> [pytest -k expression]
# quick loop while a kernel is being reworked: GPU parity tests (all, or the -k subset), then the profiled S-32G bench
tag=$1; kexpr=$2
if [ -n "$kexpr" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
else timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; fi
tail -3 gpurun_out/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_prof.sh ${tag}_prof_s32g --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end || exit 1
cut -c1-300 gpurun_out/${tag}_prof_s32g.json
python3 tools/kstats.py gpurun_out/${tag}_prof_s32g_kernel_stats.csv | head -14
