#!/bin/bash
# round 4, call 2: instruction-cache probe, the tests written since call 1, EPI-fuse A/B of configs[3]/[4]
O=gpurun_out/r4_c2; mkdir -p $O
timeout 400 bash tools/icache_probe.sh w3 c1s 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -30
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_detr_graph.py tests/test_gpu_ddp_rehearsal.py tests/test_gpu_augment.py "tests/test_gpu_resnet.py" "tests/test_gpu_detr.py" > $O/tests.log 2>&1; tail -15 $O/tests.log
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for f in 1 0 1 0; do
  MI_RESNET_EPI_FUSE=$f timeout 90 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "detr EPI_FUSE=$f"
  MI_RESNET_EPI_FUSE=$f timeout 90 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "sparseinst EPI_FUSE=$f"
done
