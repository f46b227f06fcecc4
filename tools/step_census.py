#!/usr/bin/env python
"""local: what one replayed step of a captured DETR / SparseInst step is made of.  Reads gpurun_out/trace_<tag>.csv
(tools/gpu_trace.sh <tag> --config detr|sparseinst) and the bench line in trace_<tag>.log; the timed region is the last
`steps` x ms_per_step of the timeline.  Prints dispatches and time per kernel PER STEP, the span the dispatches cover and the
idle time between them.   usage: step_census.py <tag> [top]"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.log")) if l.startswith("{")][-1]
b = json.loads(line)
steps, ms = b["steps"], b["ms_per_step"]
rows = []
for l in open(os.path.join(ROOT, "gpurun_out", f"trace_{tag}.csv")):
    p = l.rstrip("\n").split("|")
    rows.append((p[0].replace("void ", ""), int(p[1]), int(p[2])))
end = max(r[2] for r in rows)
# (the profiler stretches the step: take the region from the dispatch pattern, not from the untraced ms_per_step)
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if "adamw_multi" in n]          # once per step, its last kernel
per = collections.Counter(b - a for a, b in zip(idx[:-1], idx[1:])).most_common(1)[0][0]
k = min(steps, len(idx) - 1)
reg = rows[idx[-1] + 1 - k * per: idx[-1] + 1]
span = (reg[-1][2] - reg[0][1]) / 1e3 / k
busy = sum(r[2] - r[1] for r in reg) / 1e3 / k
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in reg:
    a = agg[n[:100]]
    a[0] += 1
    a[1] += (e - s) / 1e3
print(f"# {tag}: bench {b['value']} {b['unit']}, {ms} ms/step untraced; traced: {per} dispatches/step, span {span:.1f} us/step, "
      f"sum of dispatch durations {busy:.1f} us/step, over the last {k} steps")
print(f"# {'kernel':100s} {'n/step':>7s} {'us/step':>9s} {'avg us':>7s} {'%':>5s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:102s} {c / k:7.1f} {t / k:9.1f} {t / c:7.2f} {100 * t / k / busy:5.1f}")
small = sum(c for n, (c, t) in agg.items() if t / c < 6.0) / k
print(f"# dispatches averaging < 6 us: {small:.0f} per step")
