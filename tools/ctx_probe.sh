#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
RAW=/tmp/ctxp; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python $GRAFT_REPO_ROOT/tools/ctx_probe.py > $OUT/ctx_probe.log 2>&1)
tail -2 $OUT/ctx_probe.log
F=$(find $RAW -name '*_kernel_trace.csv' | head -1)
python - "$F" "$OUT/ctx_order.txt" "$OUT/ctx_probe.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
order = [l.rstrip("\n") for l in open(sys.argv[2])]
seqs, cur = [], None
for r in rows:
    n = r["Kernel_Name"]; grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
    if "code_polluter" in n and grid == 64:
        if cur is not None: seqs.append(cur)
        cur = []; continue
    if cur is not None:
        cur.append((n.split("(")[0].replace("void ", "")[:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
with open(sys.argv[3], "w") as f:
    for name, s in zip(order, seqs):
        f.write(name + "\n   " + " ".join(f"{n.split('<')[0][:10]}:{d:.1f}" for n, d in s) + "\n")
print(open(sys.argv[3]).read()[:6000])
PY
