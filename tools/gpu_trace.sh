#!/bin/bash
# Runs on the GPU box: per-dispatch kernel trace of a few graph-replayed steps -> gpurun_out/trace_<tag>.csv (name,start,end)
# usage: tools/gpu_trace.sh <tag>   (env such as MI355_LIB / MI_CONV_TUNE is inherited)
set -u
TAG=${1:-t}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
RAW=/tmp/trace_$TAG
rm -rf $RAW; mkdir -p $RAW
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $RAW -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline "$@" > $OUT/trace_$TAG.log 2>&1
F=$(find $RAW -name '*_kernel_trace.csv' | head -1)
python - "$F" "$OUT/trace_$TAG.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    for r in rows:
        f.write(f'{r["Kernel_Name"].split("(")[0][:90]}|{r["Start_Timestamp"]}|{r["End_Timestamp"]}|{r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))}|{r.get("Grid_Size", r.get("Grid_Size_X", ""))}\n')
print("dispatches", len(rows))
PY
tail -2 $OUT/trace_$TAG.log
