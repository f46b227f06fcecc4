#!/bin/bash
# Runs on the GPU box: kernel + memory-copy trace of 12 host-fed steps, for MI_FEED_DIRECT=0 and =1
# -> gpurun_out/h2d_trace_{0,1}.txt (summary by tools/h2d_trace_summary.py)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for d in 0 1; do
  RAW=/tmp/h2dtrace_$d; rm -rf $RAW; mkdir -p $RAW
  MI_FEED_DIRECT=$d rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $RAW -o trace -- python $GRAFT_REPO_ROOT/tools/h2d_probe.py --trace > $OUT/h2d_trace_$d.log 2>&1
  K=$(find $RAW -name '*_kernel_trace.csv' | head -1); Mc=$(find $RAW -name '*_memory_copy_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/h2d_trace_summary.py "$K" "$Mc" > $OUT/h2d_trace_$d.txt 2>&1
  tail -30 $OUT/h2d_trace_$d.txt
done
