#!/bin/bash
# SQ counters per kernel (one --pmc pass per counter set): tools/gpu_pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$TAG
mkdir -p $OUT; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/raw$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-graph > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
  rm -rf $OUT/raw$i
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/set*.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/sq_summary.csv", "w") as fo:
    names = sorted({c for v in agg.values() for c in v})
    fo.write("kernel,launches," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
        n = max(len(x) for x in v.values())
        fo.write('"%s",%d,' % (k, n) + ",".join("%.0f" % (sum(v[c]) / len(v[c])) if c in v else "" for c in names) + "\n")
print(open("$OUT/sq_summary.csv").read()[:6000])
PY
