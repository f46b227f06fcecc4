#!/usr/bin/env python
"""local: in-graph duration of EVERY command of the YOLOX-s step.  Joins gpurun_out/trace_<tag>.csv (rocprofv3 kernel trace
of graph-replayed steps, tools/gpu_trace.sh) with gpurun_out/tags.txt (tools/dump_tags.py) by walking both in launch order.
usage: trace_steps.py <tag> [--warm]   (--warm: per (kernel, shape) the fastest instance and the excess of the others)"""
import collections, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDISP = {"STREAM": 0, "LOSS_FWD": 4, "LOSS_BWD_FUSED": 2, "WGRAD_GROUP": None}


def load(tag):
    rows = []
    for l in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.csv")):
        p = l.rstrip("\n").split("|")
        rows.append((p[0].replace("void ", ""), int(p[1]), int(p[2]), p[4] if len(p) > 4 else ""))
    starts = [i for i, r in enumerate(rows) if "focus_pack" in r[0]]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][-8:]
    out = []
    for i in range(n):
        d = sorted(s[i][2] - s[i][1] for s in steps)
        out.append((steps[0][i][0], d[len(d) // 2] / 1e3, steps[0][i][3]))
    span = sorted(s[-1][2] - s[0][1] for s in steps)[len(steps) // 2] / 1e3
    return out, span


def commands(tag=None):
    cmds = []
    per_tag = os.path.join(ROOT, "gpurun_out", f"tags_{tag}.txt")
    for l in open(per_tag if tag and os.path.exists(per_tag) else os.path.join(ROOT, "gpurun_out", "tags.txt")):
        if l.startswith(("fwd", "bwd")):
            p = l.rstrip("\n").split(None, 4)
            cmds.append((p[0], p[2], p[3], p[4] if len(p) > 4 else ""))
    # the trace's step starts at FOCUS; PACK_W_BATCH + the forward MEMSET of the NEXT step close it (after the SGD kernel)
    k = next(i for i, c in enumerate(cmds) if c[1] == "FOCUS")
    return cmds[k:] + [("opt", "SGD", "sgd", "")] + cmds[:k]


def join(tag):
    disp, span = load(tag)
    cmds = commands(tag)
    fixed = sum(NDISP.get(c[1], 1) or 0 for c in cmds if c[1] != "WGRAD_GROUP")
    nwg = len(disp) - fixed
    out, i = [], 0
    for c in cmds:
        n = nwg if c[1] == "WGRAD_GROUP" else NDISP.get(c[1], 1)
        ds = disp[i:i + n]; i += n
        out.append((c, ds))
    assert i == len(disp), (i, len(disp))
    return out, span


if __name__ == "__main__":
    tag = sys.argv[1]
    rows, span = join(tag)
    tot = sum(d[1] for _, ds in rows for d in ds)
    print(f"# {tag}: step span {span:.1f} us, sum of dispatch durations {tot:.1f} us, {sum(len(ds) for _, ds in rows)} dispatches")
    fam = collections.defaultdict(lambda: [0.0, 0])
    for c, ds in rows:
        for d in ds:
            fam[c[1]][0] += d[1]; fam[c[1]][1] += 1
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"#   {k:16s} {v[0]:8.1f} us {v[1]:4d} dispatches")
    if "--warm" in sys.argv:
        # commands that run the same kernel on the same shape: the fastest instance vs the others
        groups = collections.defaultdict(list)
        for c, ds in rows:
            if c[1] in ("CONV", "BN_ACT_FWD", "BN_BWD_REDUCE", "BN_BWD_APPLY", "BN_BWD_FUSED") and ds:
                shape = re.sub(r"tile\S+|BN\d+|KC\d+|T\d+$", "", c[3]).strip()
                groups[(ds[0][0], shape)].append((ds[0][1], c[0], c[2]))
        ex = 0.0
        for (kn, shape), v in sorted(groups.items(), key=lambda kv: -(sum(x[0] for x in kv[1]) - len(kv[1]) * min(x[0] for x in kv[1]))):
            if len(v) < 2:
                continue
            mn = min(x[0] for x in v); e = sum(x[0] for x in v) - len(v) * mn; ex += e
            print(f"{kn[:44]:44s} {shape:34s} n={len(v):2d} min {mn:6.1f} excess {e:7.1f}  " + " ".join(f"{x[0]:.0f}" for x in v))
        print(f"# total excess over the fastest instance of the same (kernel, shape): {ex:.1f} us")
    else:
        for c, ds in rows:
            print(f"{c[0]} {c[1]:14s} {c[2][:70]:70s} {c[3][:44]:44s} " + " ".join(f"{d[1]:.1f}" for d in ds))
