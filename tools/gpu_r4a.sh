#!/bin/bash
# round 4, first GPU contact of the recursive parse sort: forced-route checks, GPU suite, S-32G with stage log
tag=$1
for env in "PFP_PARSE_REC=1" "PFP_PARSE_REC=1 PFP_PARSE_REC_P2=7 PFP_PARSE_REC_TILE_ROWS=5" "PFP_PARSE_REC=1 PFP_PARSE_REC_DEPTH=3 PFP_PARSE_REC_P2=3 PFP_PARSE_REC_TILE_ROWS=40"; do
  echo "== $env" >> gpurun_out/${tag}_rec_check.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/rec_check.py gpu >> gpurun_out/${tag}_rec_check.log 2>&1 || { tail -5 gpurun_out/${tag}_rec_check.log; exit 1; }
done
tail -3 gpurun_out/${tag}_rec_check.log
PFP_VERBOSE=1 timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
cut -c1-600 gpurun_out/${tag}_bench_s32g.json
grep -n "recursive\|D2 sorted\|P2 sorted\|assembl" gpurun_out/${tag}_bench_s32g.err | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_pytest.log
exit $rc
