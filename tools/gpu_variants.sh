#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_variants.sh <tag>
# A/B runs of differently tuned builds of the library (pfbwt-f_amd/lib/variants/libpfbwt_hip_<name>.so, built with other
# -DPFP_* tunables; PFBWT_HIP_LIB selects the build): ms per step and per stage of the default bench workload
tag=$1
for so in pfbwt-f_amd/lib/libpfbwt_hip.so pfbwt-f_amd/lib/variants/*.so; do
  n=$(basename $so .so)
  PFBWT_HIP_LIB=$so timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.err
  python3 - "$n" gpurun_out/${tag}_$n.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-28s %.1f ms/step  stages %s" % (sys.argv[1], d["ms_per_step"], {k: round(v, 1) for k, v in d["stage_ms"].items()}), flush=True)
except Exception as e:
    print(sys.argv[1], "failed", e, flush=True)
P
done
