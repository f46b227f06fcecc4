#!/bin/bash
# SQ counters + kernel durations of the attention kernels (tools/mha_kernel_bench.py); pattern = kernel name substring
PAT=${1:-mha_}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mha; mkdir -p $OUT; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/raw$i -o p -- python $GRAFT_REPO_ROOT/tools/mha_kernel_bench.py > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
  find $OUT/raw$i -name '*kernel_trace.csv' -exec cp {} $OUT/trace$i.csv \;
  rm -rf $OUT/raw$i
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw0 -o p -- python $GRAFT_REPO_ROOT/tools/mha_kernel_bench.py > $OUT/log0.txt 2>&1
find $OUT/raw0 -name '*kernel_trace.csv' -exec cp {} $OUT/trace0.csv \;
rm -rf $OUT/raw0
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/set*.csv")):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:44], r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v): print("   %-28s %14.0f  (n=%d)" % (c, sum(v[c]) / len(v[c]), len(v[c])))
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/trace0.csv")):
    if "$PAT" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r.get("VGPR_Count"), r.get("LDS_Block_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v); print(k, "n", len(v), "median us %.2f min %.2f" % (v[len(v) // 2], v[0]))
PY
