#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc.sh <tag> [bench args]
# HBM traffic per kernel (two counter passes, MI355X_MICROARCH.md section on FETCH_SIZE / WRITE_SIZE) -> gpurun_out/<tag>_pmc_traffic.json
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > /dev/null 2> $root/gpurun_out/${tag}_pmc_$ctr.err; echo "pmc $ctr rc=$?"
done
cd $root
python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic.json S-32G > gpurun_out/${tag}_pmc_traffic.txt 2>&1; echo "pmc summary rc=$?"
head -30 gpurun_out/${tag}_pmc_traffic.txt
