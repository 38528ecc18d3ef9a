#!/bin/bash
# p2 sweep of the recursive parse sort on S-32G (stage times only)
tag=$1; shift
for p2 in "$@"; do
  PFP_TEST_HOOKS=1 PFP_PARSE_REC_P2=$p2 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_p2_${p2}.json 2> gpurun_out/${tag}_p2_${p2}.err || { tail -5 gpurun_out/${tag}_p2_${p2}.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/${tag}_p2_${p2}.json').read()); print('p2=$p2', round(d['ms_per_step'],1), d['stage_ms'])"
done
