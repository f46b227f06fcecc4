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

# ---- BatchNorm launches: achieved bytes/s (fwd: read y [+res], write out; reduce: read dy, y; apply: read dy, y, write dy [+dres])
import re
for op, kname, nrw in (("BN_ACT_FWD", "bn_act_fwd", (1, 1)), ("BN_BWD_REDUCE", "bn_bwd_reduce", (2, 0)), ("BN_BWD_APPLY", "bn_bwd_apply", (2, 1))):
    cmds = [t for t in tags if t[2] == op]
    rows = [r for r in res[a][0] if kname in r[0]]
    if len(cmds) != len(rows):
        print(op, "count mismatch", len(cmds), len(rows)); continue
    tot_b = tot_t = 0.0
    by = {}
    for t, r in zip(cmds, rows):
        m = re.search(r"count(\d+) C(\d+) res(\d)", t[4])
        if not m: continue
        cnt, Cc, rs = int(m.group(1)), int(m.group(2)), int(m.group(3))
        b = cnt * Cc * 2 * (nrw[0] + nrw[1] + rs)
        tot_b += b; tot_t += r[1]
        key = (cnt // 16, Cc)
        by.setdefault(key, [0, 0.0, 0.0]); by[key][0] += 1; by[key][1] += b; by[key][2] += r[1]
    print(f"{op}: {tot_t:.1f} us, {tot_b / 1e6:.0f} MB, {tot_b / tot_t / 1e6:.2f} TB/s")
    for key in sorted(by, reverse=True):
        n, b, tt = by[key]
        print(f"    HW{key[0]:6d} C{key[1]:4d} x{n:2d}: {tt / n:6.1f} us  {b / tt / 1e6:5.2f} TB/s")
