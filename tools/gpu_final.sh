#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_final.sh <tag>
# the evidence set of a round: default bench line, kernel-trace summary of the same command, PMC traffic passes, the other
# single-GPU workloads (S-chr22 = configs[1], S-3G = configs[2]) with their kernel traces -- everything under gpurun_out/<tag>_*
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
tools/gpu_prof.sh ${tag}_prof_s32g --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end; echo "prof rc=$?"
mv gpurun_out/${tag}_prof_s32g_kernel_stats.csv gpurun_out/${tag}_bench_s32g_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $root/gpurun_out/${tag}_pmc_$ctr.err; echo "pmc $ctr rc=$?"
done
cd $root
python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic_s32g.json S-32G > gpurun_out/${tag}_pmc_traffic_s32g.txt 2>&1; echo "pmc summary rc=$?"
for wl in S-chr22 S-3G; do      # HBM traffic of the other workloads' kernels (their bench lines look profiles/pmc_traffic_<workload>.json up)
  cd /tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python3 $root/bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $root/gpurun_out/${tag}_pmc_${wl}_$ctr.err; echo "pmc $wl $ctr rc=$?"
  done
  cd $root
  python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic_$wl.json $wl > gpurun_out/${tag}_pmc_traffic_$wl.txt 2>&1; echo "pmc summary $wl rc=$?"
  cp gpurun_out/${tag}_pmc_traffic_$wl.json profiles/pmc_traffic_$wl.json
done
cp gpurun_out/${tag}_pmc_traffic_s32g.json profiles/pmc_traffic_latest.json
python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_bench_s32g.json 2> gpurun_out/${tag}_bench_s32g.err; echo "default bench rc=$?"
python bench.py --workload S-chr22 --steps 20 --warmup 2 > gpurun_out/${tag}_bench_chr22.json 2> gpurun_out/${tag}_bench_chr22.err; echo "chr22 rc=$?"
tools/gpu_prof.sh ${tag}_prof_chr22 --workload S-chr22 --steps 9 --warmup 1 --no-cpu-baseline --no-end-to-end
mv gpurun_out/${tag}_prof_chr22_kernel_stats.csv gpurun_out/${tag}_bench_chr22_kernel_stats.csv
python bench.py --workload S-3G --steps 3 --warmup 1 > gpurun_out/${tag}_bench_s3g.json 2> gpurun_out/${tag}_bench_s3g.err; echo "S-3G rc=$?"
tools/gpu_prof.sh ${tag}_prof_s3g --workload S-3G --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end
mv gpurun_out/${tag}_prof_s3g_kernel_stats.csv gpurun_out/${tag}_bench_s3g_kernel_stats.csv
cp profiles/pmc_traffic_S-chr22.json profiles/pmc_traffic_S-3G.json gpurun_out/ 2>/dev/null
