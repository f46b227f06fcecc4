"""local: what the weight-gradient side queue does to the step, from in-graph traces (tools/ab_trace.sh):
python tools/side_queue_trace_summary.py <default tag> <tag> [<tag> ...]
Per trace, over the last complete step (Focus launch to Focus launch): span, busy time of the chain kernels and of the
weight-gradient kernels, time both run at once, and the duration of every kernel family against the default trace."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag):
    rows = []
    for ln in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.csv")):
        p = ln.rstrip("\n").split("|")
        rows.append((p[0], int(p[1]), int(p[2])))
    rows.sort(key=lambda r: r[1])
    foc = [i for i, r in enumerate(rows) if "focus" in r[0]]
    # the last Focus-to-Focus interval that holds a whole step (later ones belong to bench.py's per-command timing)
    a, b = [(x, y) for x, y in zip(foc[:-1], foc[1:]) if y - x >= 200][-1]
    return rows[a:b], rows[b][1] - rows[a][1]


def fam(n):
    n = n.replace("void ", "")
    return n.split("<")[0][:34]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap(a, b):
    ev = [(s, 1, 0) for s, e in a] + [(e, -1, 0) for s, e in a] + [(s, 0, 1) for s, e in b] + [(e, 0, -1) for s, e in b]
    ev.sort()
    na = nb = 0
    last, tot = None, 0
    for t, da, db in ev:
        if last is not None and na > 0 and nb > 0:
            tot += t - last
        na += da
        nb += db
        last = t
    return tot


base = None
for tag in sys.argv[1:]:
    rows, span = load(tag)
    wg = [(s, e) for n, s, e in rows if "wgrad" in n]
    ch = [(s, e) for n, s, e in rows if "wgrad" not in n]
    f = collections.OrderedDict()
    for n, s, e in rows:
        k = fam(n)
        c = f.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e - s
    print(f"\n== {tag}: step span {span / 1e3:.1f} us, {len(rows)} dispatches; chain busy (union) {union(ch) / 1e3:.1f} us, "
          f"sum of chain kernels {sum(e - s for s, e in ch) / 1e3:.1f} us; weight-gradient kernels: {len(wg)} launches, "
          f"sum {sum(e - s for s, e in wg) / 1e3:.1f} us, busy (union) {union(wg) / 1e3:.1f} us; both at once {overlap(ch, wg) / 1e3:.1f} us")
    if base is None:
        base = f
    else:
        for k, (c, t) in f.items():
            if k in base and abs(t - base[k][1]) > 5000:
                print(f"   {k:36s} {c:3d} launches {base[k][1] / 1e3:8.1f} -> {t / 1e3:8.1f} us ({(t - base[k][1]) / 1e3:+.1f})")
