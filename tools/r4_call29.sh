#!/bin/bash
cd $GRAFT_REPO_ROOT
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for r in 1 2 3; do for f in 1 0; do
  MI_DETR_MATCH_LEVELS=$f timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr MATCH_LEVELS=$f"
done; done
