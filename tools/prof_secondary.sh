#!/bin/bash
# rocprofv3 --kernel-trace --stats of a secondary bench: CFG=detr|sparseinst bash tools/prof_secondary.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${CFG:-detr}; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o trace -- python $GRAFT_REPO_ROOT/bench.py --config ${CFG:-detr} --steps 5 --warmup 2 > $OUT/bench.log 2>&1
find $OUT/raw -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/raw
tail -2 $OUT/bench.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "over 7 steps ->", tot/1e6/7, "ms/step; kernels launched", sum(int(r['Calls']) for r in rows), "per step", sum(int(r['Calls']) for r in rows)/7)
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f}us tot {float(r['TotalDurationNs'])/1e6:8.2f}ms {float(r['Percentage']):5.2f}%")
PY
