#!/bin/bash
O=gpurun_out/r4_c12; mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_step.py tests/test_gpu_parity_bench.py > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
bash tools/abenv.sh "MI_PACK_ASYNC=0" "MI_PACK_ASYNC=1" 40
bash tools/abenv.sh "MI_PACK_ASYNC=0" "MI_PACK_ASYNC=1" 40
