#!/usr/bin/env python
"""local: per-command difference of two in-graph traces (tools/ab_trace.sh): trace_diff.py <tagA> <tagB> [fwd|bwd]
Commands are matched by tag; a BatchNorm launch of A that B folded into a consumer shows up as 'only in A'."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import trace_steps as T


def table(tag):
    rows, span = T.join(tag)
    return collections.OrderedDict(((c[0], c[2]), (c[1], sum(d[1] for d in ds))) for c, ds in rows), span


if __name__ == "__main__":
    a, sa = table(sys.argv[1])
    b, sb = table(sys.argv[2])
    which = sys.argv[3] if len(sys.argv) > 3 else None
    print(f"# span {sa:.1f} -> {sb:.1f} us")
    only_a = only_b = both = 0.0
    for k, (op, va) in a.items():
        if which and k[0] != which:
            continue
        if k in b:
            vb = b[k][1]
            both += vb - va
            if abs(vb - va) >= 1.0:
                print(f"{k[0]} {op:14s} {k[1][:64]:64s} {va:7.1f} -> {vb:7.1f}  ({vb - va:+.1f})")
        else:
            only_a += va
            print(f"{k[0]} {op:14s} {k[1][:64]:64s} {va:7.1f} -> (gone)")
    for k, (op, vb) in b.items():
        if (not which or k[0] == which) and k not in a:
            only_b += vb
            print(f"{k[0]} {op:14s} {k[1][:64]:64s}    (new) -> {vb:7.1f}")
    print(f"# common commands {both:+.1f} us, only in A {only_a:.1f} us, only in B {only_b:.1f} us")
