#!/bin/bash
# GPU box: tools/icache_probe.py under rocprofv3 --kernel-trace -> gpurun_out/icache_<which>.txt (sequence tables)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for which in ${@:-w3 c1s}; do
RAW=/tmp/icp_$which; rm -rf $RAW; mkdir -p $RAW
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python $GRAFT_REPO_ROOT/tools/icache_probe.py $which > $OUT/icache_$which.log 2>&1)
F=$(find $RAW -name '*_kernel_trace.csv' | head -1)
python - "$F" "$OUT/icache_$which.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# replays are separated by a ONE-block launch of the polluter kernel (grid 64); keep the last replay of every sequence
seqs, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
    if "code_polluter" in n and grid == 64:
        if cur: seqs.append(cur)
        cur = []
        continue
    if cur is not None:
        cur.append((n.split("(")[0].replace("void ", "")[:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
if cur: seqs.append(cur)
with open(sys.argv[2], "w") as f:
    for i, s in enumerate(seqs):
        if i % 4 == 3:     # last of the four replays
            conv = [d for n, d in s if "w3_" in n or "c1s_" in n or "conv_igemm" in n]
            other = {}
            for n, d in s:
                if not ("w3_" in n or "c1s_" in n or "conv_igemm" in n): other.setdefault(n, []).append(d)
            f.write(f"seq {i // 4}: conv " + " ".join(f"{d:.1f}" for d in conv) + " | " + " ; ".join(f"{n} avg {sum(v) / len(v):.1f}" for n, v in other.items()) + "\n")
print(open(sys.argv[2]).read())
PY
tail -2 $OUT/icache_$which.log
done
