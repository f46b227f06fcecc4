#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the K = 32 / 64 weight-stationary launches: tools/w3_k32_pmc.sh <tag>
set -u
TAG=${1:-w3}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp
i=0
for SET in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  ITERS=2 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/raw$i -o p -- python $GRAFT_REPO_ROOT/tools/w3_k32_probe.py > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
  rm -rf $OUT/raw$i
  tail -2 $OUT/log$i.txt
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/set*.csv")):
    for r in csv.DictReader(open(f)):
        if "w3_kernel" not in r["Kernel_Name"]: continue
        agg[(r["Kernel_Name"][:48], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
