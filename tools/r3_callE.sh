#!/bin/bash
O=gpurun_out/final3; mkdir -p $O
timeout 60 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2> $O/detr.err | tail -1 > $O/bench_detr.json
timeout 60 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2> $O/si.err | tail -1 > $O/bench_sparseinst.json
timeout 110 python -m pytest tests/test_gpu_conv1x1_stream.py tests/test_gpu_conv3x3_ws.py tests/test_gpu_conv_bn_fused.py tests/test_gpu_dwconv.py tests/test_gpu_augment.py tests/test_eval_rle.py tests/test_ref_nms.py -x -q -m gpu > $O/tests2.log 2>&1
tail -2 $O/tests2.log
python - <<'PY'
import json
for n in ("bench_detr","bench_sparseinst"):
    d=json.load(open(f"gpurun_out/final3/{n}.json")); print(n, d["value"], d["ms_per_step"])
PY
