#!/bin/bash
O=gpurun_out/r4_c6; mkdir -p $O
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_sparseinst.py tests/test_gpu_detr_graph.py tests/test_gpu_resnet.py tests/test_gpu_detr_meta.py > $O/tests.log 2>&1; tail -30 $O/tests.log | cut -c1-300
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for g in "" "--no-graph" "" "--no-graph"; do timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline $g 2>$O/si_err.log | val "sparseinst $g"; done
timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr"
