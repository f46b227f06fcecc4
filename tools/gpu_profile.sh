#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command -> gpurun_out/prof_<tag>/
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_under_prof.log" 2>&1
find "$OUT/raw" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/raw" -name '*_kernel_trace.csv' -exec sh -c 'head -1 "$1" > "$2/kernel_trace_head.csv"' _ {} "$OUT" \;
rm -rf "$OUT/raw"
head -40 "$OUT/kernel_stats.csv"
