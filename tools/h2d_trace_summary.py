"""summary of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/h2d_probe.py --trace: per fed step (Focus packer
start to the next one) the period, the kernel-busy time, the gaps > 2 us between consecutive dispatches with the kernels on
both sides, and where the host -> device copies sit relative to the step."""
import csv
import sys

k = list(csv.DictReader(open(sys.argv[1])))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
m = list(csv.DictReader(open(sys.argv[2]))) if len(sys.argv) > 2 and sys.argv[2] else []
name = lambda r: r["Kernel_Name"].split("(")[0][:60]
foc = [i for i, r in enumerate(k) if "focus" in r["Kernel_Name"]]
print("dispatches", len(k), "focus launches", len(foc), "memory copies", len(m))
if m:
    print("copy columns:", list(m[0].keys()))
cps = []
for r in m:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    cps.append((s, e, r.get("Direction", r.get("Kind", "")), r.get("Size", r.get("Bytes", ""))))
cps.sort()
for a, b in zip(foc[-7:-1], foc[-6:]):
    rows = k[a:b]
    t0, t1 = int(rows[0]["Start_Timestamp"]), int(k[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    print(f"\nstep: period {(t1 - t0) / 1e3:.1f} us, {len(rows)} dispatches, busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
    ext = rows + [k[b]]
    gaps = []
    for x, y in zip(ext[:-1], ext[1:]):
        g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        if g > 2000:
            gaps.append((g, name(x), name(y), (int(x["End_Timestamp"]) - t0) / 1e3))
    for g, x, y, at in gaps:
        print(f"   gap {g / 1e3:7.1f} us at +{at:7.1f} us: {x} -> {y}")
    for s, e, d, sz in cps:
        if s < t1 and e > t0:
            print(f"   copy {d} {sz} B: +{(s - t0) / 1e3:.1f} .. +{(e - t0) / 1e3:.1f} us ({(e - s) / 1e3:.1f} us)")
    oth = [name(r) for r in rows if "elementwise" in r["Kernel_Name"] or "copy" in r["Kernel_Name"].lower()]
    if oth:
        print("   non-plan kernels:", oth)
