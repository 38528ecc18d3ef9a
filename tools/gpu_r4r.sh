#!/bin/bash
# FASTA records counted into the sequence hint: the end-to-end legs take the cooperative kernel and the column order too
tag=$1
timeout -k 10 900 python -m pytest tests/test_fasta_ingest.py tests/test_host_cli.py -m gpu -x -q > gpurun_out/${tag}_pytest_fasta.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest_fasta.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest_fasta.log
timeout -k 10 1000 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err || { tail -20 gpurun_out/${tag}_bench_s32g.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_s32g.json').read()); print(round(d['ms_per_step'],1), d['stage_ms'])
e=d['end_to_end_fasta']
for k in ('cold','warm','warm_streamed'): print(k, round(e[k]['ms']), round(e[k]['value'],1), 'parse', round(e[k]['parse_ms'],1), 'ingest', round(e[k]['ingest_ms']))
print('match', e.get('match_device_outputs'))"
