#!/usr/bin/env python
"""local: join gpurun_out/trace_<tag>.csv (per-dispatch kernel trace of graph-replayed steps) with gpurun_out/tags.txt:
in-graph duration of every conv launch, averaged over the last steps.  usage: trace_table.py tagA [tagB]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(tag):
    rows = []
    for l in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.csv")):
        p = l.rstrip("\n").split("|")
        rows.append((p[0], int(p[1]), int(p[2])))
    starts = [i for i, r in enumerate(rows) if "focus_pack" in r[0]]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][-4:]
    out = []
    for i in range(n):
        name = steps[0][i][0]
        dur = sum(s[i][2] - s[i][1] for s in steps) / len(steps) / 1e3
        gap = sum((s[i][1] - s[i - 1][2]) if i else 0 for s in steps) / len(steps) / 1e3
        out.append((name, dur, gap))
    span = sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3
    return out, span
tags = [l.split(None, 4) for l in open(os.path.join(ROOT, "gpurun_out", "tags.txt")) if l.startswith(("fwd", "bwd"))]
convs = [t for t in tags if t[2] == "CONV"]
res = {}
for tg in sys.argv[1:]:
    rows, span = load(tg)
    cv = [r for r in rows if "conv_igemm" in r[0]]
    assert len(cv) == len(convs), (len(cv), len(convs))
    res[tg] = (rows, cv, span)
    cls = collections.defaultdict(float); gaps = 0.0
    for name, dur, gap in rows:
        key = name.split("<")[0].replace("void ", "")
        cls[key] += dur; gaps += gap
    print(f"== {tg}: focus->last kernel span {span:.1f} us, sum of kernel durations {sum(r[1] for r in rows):.1f} us, gaps {gaps:.1f} us, {len(rows)} dispatches")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1])[:14]:
        print(f"   {k:40s} {v:8.1f} us")
a = sys.argv[1]
b = sys.argv[2] if len(sys.argv) > 2 else None
tot = [0.0, 0.0]
for i, t in enumerate(convs):
    da = res[a][1][i][1]
    line = f"{t[0]} {t[3]:38s} {t[4].strip() if len(t) > 4 else '':60s} {da:7.1f}"
    tot[0] += da
    if b:
        db = res[b][1][i][1]; tot[1] += db
        line += f" {db:7.1f} {db - da:+6.1f}  {res[b][1][i][0][-28:]}"
    print(line)
print("conv total", tot)
