#!/bin/bash
O=gpurun_out/r4_c5; mkdir -p $O
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_detr_graph.py "tests/test_gpu_resnet.py" > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout 100 python tools/dump_tags.py > gpurun_out/tags.txt 2>/dev/null
MI_SIMOTA_V2=1 timeout 200 bash tools/gpu_trace.sh r4v2
MI_SIMOTA_V2=0 timeout 200 bash tools/gpu_trace.sh r4v1
for f in 1 0; do MI_RESNET_EPI_FUSE=$f timeout 90 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('detr EPI_FUSE=$f', d['value'], d['ms_per_step'])"; done
