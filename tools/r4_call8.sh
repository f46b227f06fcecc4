#!/bin/bash
N=6 timeout 400 python tools/detr_graph_check.py 2>&1 | grep -v "Warning\|warn\|detach\|print(" | tail -8 | cut -c1-300
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for g in "" "--no-graph" "" "--no-graph"; do timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline $g 2>/dev/null | val "sparseinst $g"; done
