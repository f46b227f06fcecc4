#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ws; mkdir -p $OUT; cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/raw$i -o p -- python $GRAFT_REPO_ROOT/tools/ws_pmc.py > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
  find $OUT/raw$i -name '*kernel_trace.csv' -exec cp {} $OUT/trace$i.csv \;
  rm -rf $OUT/raw$i
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/set*.csv")):
    for r in csv.DictReader(open(f)):
        if "w3_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v): print("   %-28s %14.0f  (n=%d)" % (c, sum(v[c]) / len(v[c]), len(v[c])))
for f in sorted(glob.glob("$OUT/trace*.csv"))[:1]:
    for r in csv.DictReader(open(f)):
        if "w3_kernel" in r["Kernel_Name"]: print("dur us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "vgpr", r.get("VGPR_Count"), "agpr", r.get("Accum_VGPR_Count"), "lds", r.get("LDS_Block_Size"))
PY
