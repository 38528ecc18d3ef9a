#!/bin/bash
# usage (GPU box, repo root): tools/gpu_suite.sh <tag> [pytest args...]   -- the GPU suite, log under gpurun_out/
tag=$1; shift
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 "$@" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -22 gpurun_out/${tag}_pytest.log
exit $rc
