#!/bin/bash
# cheaper counter RNG for the dropout masks: DETR tests + same-box A/B against the previous build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c30; mkdir -p $O
timeout 600 python -m pytest -q -m gpu tests/test_gpu_detr.py tests/test_gpu_detr_meta.py tests/test_gpu_detr_graph.py > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-250
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for r in 1 2; do for f in libmi355det.so libmi355det_prev.so; do
  MI355_LIB=yolov7_d2_amd/$f timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr $f"
done; done
