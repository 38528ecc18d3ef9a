#!/bin/bash
# forced-route checks of the recursive parse sort, S-32G bench line (stage times), optionally the GPU suite
tag=$1; suite=$2
for env in "PFP_PARSE_REC=1" "PFP_PARSE_REC=1 PFP_PARSE_REC_P2=7 PFP_PARSE_REC_TILE_ROWS=5" "PFP_PARSE_REC=1 PFP_PARSE_REC_DEPTH=3 PFP_PARSE_REC_P2=3 PFP_PARSE_REC_TILE_ROWS=40"; do
  echo "== $env" >> gpurun_out/${tag}_rec_check.log
  env PFP_TEST_HOOKS=1 $env timeout -k 10 300 python tools/rec_check.py gpu >> gpurun_out/${tag}_rec_check.log 2>&1 || { tail -5 gpurun_out/${tag}_rec_check.log; exit 1; }
done
grep -c " ok" gpurun_out/${tag}_rec_check.log
timeout -k 10 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'], d['roofline']['kernel'], round(d['roofline']['frac'],3))"
if [ -n "$suite" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
  tail -3 gpurun_out/${tag}_pytest.log
  exit $rc
fi
