#!/bin/bash
# counters of one kernel family inside a secondary config's eager step, per dispatch:  CFG=detr PAT=wgrad2_group tools/pmc_cfg.sh <tag>
# passes (alone, as MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, then two SQ sets -> gpurun_out/pmccfg_<tag>/summary.txt
set -u
TAG=${1:-t}; CFG=${CFG:-detr}; PAT=${PAT:-wgrad2_group}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmccfg_$TAG
mkdir -p $OUT; cd /tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/raw$i -o p -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 1 --warmup 2 --no-cpu-baseline --no-graph > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name '*counter_collection.csv' -exec cp {} $OUT/set$i.csv \;
  rm -rf $OUT/raw$i
done
python - "$OUT" "$PAT" <<'PY'
import csv, collections, glob, sys
out, pat = sys.argv[1], sys.argv[2]
per = collections.defaultdict(dict)      # dispatch id -> counter -> value
name = {}
for f in sorted(glob.glob(out + "/set*.csv")):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        # dispatches of the family in launch order; the passes run the same program: the k-th match of every pass is the same launch
        key = (r["Counter_Name"],)
        seen[key] += 1
        k = seen[key]
        per[k][r["Counter_Name"]] = float(r["Counter_Value"])
        name[k] = r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "")
n = max(per) if per else 0
last = n // 3                      # 3 steps ran (2 warm-up + 1): keep the last one
with open(out + "/summary.txt", "w") as fo:
    for k in range(n - last + 1, n + 1):
        d = per[k]
        fetch, write = 2 * d.get("FETCH_SIZE", 0) * 1024 / 1e6, d.get("WRITE_SIZE", 0) * 1024 / 1e6     # KB units; gfx950: FETCH_SIZE x2
        busy = d.get("SQ_BUSY_CYCLES", 0)
        line = (f"{k:4d} {name[k]:80s} fetch {fetch:8.1f} MB write {write:7.1f} MB | GUI {d.get('GRBM_GUI_ACTIVE', 0):9.0f} mfma_busy/busy "
                f"{d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy if busy else 0:5.2f} wait_any/wave_cycles {d.get('SQ_WAIT_ANY', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1):5.2f} "
                f"insts valu {d.get('SQ_INSTS_VALU', 0):9.0f} mfma {d.get('SQ_INSTS_MFMA', 0):8.0f} lds {d.get('SQ_INSTS_LDS', 0):8.0f} bank_conflict/lds_active "
                f"{d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.2f}")
        fo.write(line + "\n")
        print(line)
PY
