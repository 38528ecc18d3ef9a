#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_final.sh <tag>
# the evidence set of a round: default bench line, kernel-trace summary of the same command, PMC traffic passes, the other
# single-GPU workloads (S-chr22 = configs[1], S-3G = configs[2]) with their kernel traces -- everything under gpurun_out/<tag>_*
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err; echo "default bench rc=$?"
tools/gpu_prof.sh ${tag}_prof_s32g --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end; echo "prof rc=$?"
mv gpurun_out/${tag}_prof_s32g_kernel_stats.csv gpurun_out/${tag}_bench_s32g_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $root/gpurun_out/${tag}_pmc_$ctr.err; echo "pmc $ctr rc=$?"
done
cd $root
python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic_s32g.json S-32G > gpurun_out/${tag}_pmc_traffic_s32g.txt 2>&1; echo "pmc summary rc=$?"
python bench.py --workload S-chr22 --steps 20 --warmup 2 > gpurun_out/${tag}_bench_chr22.json 2> gpurun_out/${tag}_bench_chr22.err; echo "chr22 rc=$?"
tools/gpu_prof.sh ${tag}_prof_chr22 --workload S-chr22 --steps 9 --warmup 1 --no-cpu-baseline --no-end-to-end
mv gpurun_out/${tag}_prof_chr22_kernel_stats.csv gpurun_out/${tag}_bench_chr22_kernel_stats.csv
python bench.py --workload S-3G --steps 3 --warmup 1 > gpurun_out/${tag}_bench_s3g.json 2> gpurun_out/${tag}_bench_s3g.err; echo "S-3G rc=$?"
