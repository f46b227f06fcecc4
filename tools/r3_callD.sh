#!/bin/bash
# last verification of the round: the YOLOX-side GPU tests that exercise the tile convolution kernel (its epilogue gained
# MI_CONV_RELU this round), then the headline bench
O=gpurun_out/final3; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_ops_boundary.py tests/test_gpu_parity_bench.py tests/test_gpu_widths.py tests/test_gpu_step.py tests/test_gpu_bifpn.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 100 python bench.py --no-cpu-baseline --no-h2d 2> $O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/final3/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
