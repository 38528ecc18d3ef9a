#!/bin/bash
# modulus of the dictionary's level-2 parse on the card (sized on the host before: 16)
tag=$1
for p2 in 8 12 24 32; do
  PFP_TEST_HOOKS=1 PFP_DICT_REC_P2=$p2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_p2_$p2.json 2> gpurun_out/${tag}_p2_$p2.err || { tail -5 gpurun_out/${tag}_p2_$p2.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/${tag}_p2_$p2.json').read()); print($p2, round(d['ms_per_step'],1), d['stage_ms'])"
done
