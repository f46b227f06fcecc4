#!/bin/bash
O=gpurun_out/r4_c10; mkdir -p $O
timeout 1500 python -m pytest -x -q -m gpu -s tests/test_gpu_detr_meta.py -k "forward_state_pinned" > $O/pinned.log 2>&1; grep -v "Warning\|warn" $O/pinned.log | tail -8 | cut -c1-400
bash tools/r4_call11.sh
