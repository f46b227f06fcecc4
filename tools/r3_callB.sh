#!/bin/bash
# bottleneck block as one autograd node: parity subset, then the two secondary benches A/B in one call (same box)
O=gpurun_out/blockfn; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_sparseinst.py tests/test_gpu_detr_graph.py tests/test_gpu_detr_meta.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
for f in 1 0; do
  MI_RESNET_BLOCK_FN=$f timeout 120 python bench.py --config detr --steps 20 --warmup 5 2> $O/detr_$f.err | tail -1 > $O/detr_$f.json
  MI_RESNET_BLOCK_FN=$f timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 2> $O/si_$f.err | tail -1 > $O/si_$f.json
done
python - <<'PY'
import json
for n in ("detr_1","detr_0","si_1","si_0"):
    try:
        d=json.load(open(f"gpurun_out/blockfn/{n}.json")); print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
