#!/bin/bash
# reproducibility of the captured steps' final loss, run to run, at the shipped split-K floor (2) and at 1 / 4
cd $GRAFT_REPO_ROOT
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for r in 1 2 3 4; do for f in 2 1; do
  MI_WG_MIN_TILES=$f timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "sparseinst MIN_TILES=$f"
done; done
for r in 1 2 3; do
  timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr (default)"
done
