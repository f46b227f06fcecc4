#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0 1; do MI_CONV_BN_FUSE=$v TOP=400 timeout 600 python tools/layer_table.py > gpurun_out/lt_fuse$v.txt 2>&1; done
python - <<'PY'
import re
def load(f):
    rows = {}
    for l in open(f):
        m = re.match(r"(fwd|bwd) (\S+)\s+(\S+)\s+([\d.]+)us", l)
        if m: rows[m.group(2)] = (m.group(3), float(m.group(4)))
    return rows
a, b = load("gpurun_out/lt_fuse0.txt"), load("gpurun_out/lt_fuse1.txt")
tot0 = tot1 = 0
for tag, (op, us) in sorted(b.items(), key=lambda kv: -kv[1][1]):
    if ".bnact" in tag and op == "CONV_GROUP":
        parts = tag.split("+")
        n = len(parts) // 2
        conv_tag, bn_tag = "+".join(parts[:n]), "+".join(parts[n:])
        u0 = a.get(conv_tag, (None, 0))[1] + a.get(bn_tag, (None, 0))[1]
        tot0 += u0; tot1 += us
        print(f"{tag[:60]:60s} two {u0:7.1f} = {a.get(conv_tag, (None, 0))[1]:6.1f} + {a.get(bn_tag, (None, 0))[1]:6.1f}   fused {us:7.1f}  {'WIN' if us < u0 else 'lose'}")
print("total two", tot0, "fused", tot1)
PY
