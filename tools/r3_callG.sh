#!/bin/bash
O=gpurun_out/final4; mkdir -p $O
timeout 70 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_detr_graph.py tests/test_gpu_sparseinst.py -x -q -m gpu > $O/tests.log 2>&1
tail -2 $O/tests.log
timeout 25 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2> $O/detr.err | tail -1 > $O/bench_detr.json
timeout 25 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2> $O/si.err | tail -1 > $O/bench_sparseinst.json
python - <<'PY'
import json
for n in ("bench_detr","bench_sparseinst"):
    try:
        d=json.load(open(f"gpurun_out/final4/{n}.json")); print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
