#!/bin/bash
# kernel resource table of one translation unit: tools/kres.sh <file.hip> [extra hipcc flags]
# (name, VGPRs, AGPRs, scratch bytes / lane, occupancy, SGPR / VGPR spills)
cd "$(dirname "$0")/../yolov7_d2_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -Wno-inline-asm \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres_$$.o 2>&1 | python3 -c '
import sys, re
cur = {}
rows = []
for ln in sys.stdin:
    m = re.search(r"remark: +Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: +([A-Za-z /\[\]]+): (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    print("%-70s v%-4d a%-4d scratch %-4d occ %d  sspill %-3d vspill %d" % (r["name"][:70], r.get("VGPRs", -1), r.get("AGPRs", -1),
          r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1)))
'
rm -f /tmp/kres_$$.o
