#!/bin/bash
# PMC counters of the conv kernels of selected layers: tools/pmc_conv.sh <tag> "<counters>" layer...
set -u
TAG=$1; shift; PMC=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/raw -o p -- python $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" > $OUT/log.txt 2>&1
find $OUT/raw -name '*counter_collection.csv' -exec cp {} $OUT/counters.csv \;
rm -rf $OUT/raw
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "conv_igemm" not in k: continue
    agg[(k[:60], r["Grid_Size"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n", len(next(iter(v.values()))))
PY
