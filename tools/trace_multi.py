#!/usr/bin/env python
"""local: per-conv in-graph durations for several traced policies side by side; best-of summary"""
import os, sys, re
sys.argv = [sys.argv[0]] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import collections
def load(tag):
    rows = []
    for l in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.csv")):
        p = l.rstrip("\n").split("|")
        rows.append((p[0], int(p[1]), int(p[2]), p[3]))
    starts = [i for i, r in enumerate(rows) if "focus_pack" in r[0]]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][-4:]
    out = []
    for i in range(n):
        out.append((steps[0][i][0], sum(s[i][2] - s[i][1] for s in steps) / len(steps) / 1e3, steps[0][i][3]))
    span = sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3
    return out, span
tags = [l.split(None, 4) for l in open(os.path.join(ROOT, "gpurun_out", "tags.txt")) if l.startswith(("fwd", "bwd"))]
convs = [t for t in tags if t[2] == "CONV"]
names = sys.argv[1:]
data = {}
for tg in names:
    rows, span = load(tg)
    cv = [r for r in rows if "conv_igemm" in r[0]]
    data[tg] = cv
    print(f"{tg}: span {span:.1f} us  conv {sum(r[1] for r in cv):.1f} us  other {sum(r[1] for r in rows) - sum(r[1] for r in cv):.1f}")
best_tot = 0.0
wins = collections.Counter()
for i, t in enumerate(convs):
    shape = re.sub(r" tile.*", "", t[4].strip()) if len(t) > 4 else ""
    vals = [data[tg][i][1] for tg in names]
    b = min(range(len(vals)), key=lambda j: vals[j])
    best_tot += vals[b]; wins[names[b]] += 1
    def cfg(r):
        m = re.search(r"<(\d+), (\d+), \d+, \d+, \d+, \d+, (\d+)>", r[0])
        return f"{m.group(1)}/{m.group(2)}/{'m' if m.group(3) == '0' else '1'}/{int(r[2]) // 1024}k" if m else "?"
    print(f"{t[0]} {t[3]:36s} {shape:34s} " + " ".join(f"{v:6.1f}" for v in vals) + f"  best {names[b]:3s} " +
          " ".join(cfg(data[tg][i]) for tg in names))
print("best-of total", round(best_tot, 1), dict(wins))
