#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c27; mkdir -p $O
timeout 600 python -m pytest -q -m gpu tests/test_gpu_detr.py tests/test_gpu_detr_meta.py -x > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for r in 1 2; do timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr"; done
STEPS=40 ROUNDS=2 bash tools/abn.sh "MI_X=0" "MI_BN_FUSED_ITEMS=2" "MI_BN_FUSED_ITEMS=8" "MI_BN_FUSED_ITEMS=3"
