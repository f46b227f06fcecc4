#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_detr.py tests/test_gpu_detr_meta.py tests/test_ops_boundary.py -q 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -20
timeout 600 python bench.py --config detr 2>/dev/null | tail -1 > gpurun_out/bench_detr_r03b.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detr_r03b.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:600])
PY
