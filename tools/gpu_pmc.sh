#!/bin/bash
# HBM traffic counters per kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes): tools/gpu_pmc.sh <tag>
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/raw_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/log_$C.txt 2>&1
  find $OUT/raw_$C -name '*counter_collection.csv' -exec cp {} $OUT/$C.csv \;
  rm -rf $OUT/raw_$C
done
python - <<PY
import csv, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s.csv" % c)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = (sum(v), len(v))
rows = []
for k, d in res.items():
    f, n = d.get("FETCH_SIZE", (0, 1)); w, _ = d.get("WRITE_SIZE", (0, 1))
    # rocprofv3 units: KB; gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section)
    rows.append((2 * f * 1024 + w * 1024, k, n, 2 * f * 1024 / n, w * 1024 / n))
rows.sort(reverse=True)
with open("$OUT/hbm_traffic.csv", "w") as fo:
    fo.write("kernel,launches,fetch_bytes_per_launch_x2corrected,write_bytes_per_launch\n")
    for t, k, n, f, w in rows:
        fo.write('"%s",%d,%.0f,%.0f\n' % (k, n, f, w))
for t, k, n, f, w in rows[:14]:
    print("%-80s n=%5d fetch %8.2f MB write %8.2f MB per launch" % (k[:80], n, f / 1e6, w / 1e6))
PY
