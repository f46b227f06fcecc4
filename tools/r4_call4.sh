#!/bin/bash
O=gpurun_out/r4_c4; mkdir -p $O
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_step.py tests/test_gpu_parity_bench.py -k "simota or loss or losses" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python -m pytest -x -q -m gpu "tests/test_gpu_resnet.py" -k epilogue > $O/epi.log 2>&1; tail -4 $O/epi.log
bash tools/abenv.sh "MI_SIMOTA_V2=0" "MI_SIMOTA_V2=1" 40
