#!/bin/bash
# usage (GPU box, repo root): tools/gpu_trace.sh <tag> [bench args...]  -- rocprofv3 kernel trace WITH per-dispatch timestamps
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $root/bench.py "$@" > $root/gpurun_out/$tag.json 2> $root/gpurun_out/$tag.err
rc=$?
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/gpurun_out/${tag}_kernel_stats.csv
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 $root/tools/trace_compact.py $f $root/gpurun_out/${tag}_timeline.csv
cd $root
exit $rc
