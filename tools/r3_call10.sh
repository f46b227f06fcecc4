#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv3x3_ws.py tests/test_gpu_conv1x1_stream.py -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_parity_bench.py tests/test_gpu_widths.py -x -q 2>&1 | tail -6
STEPS=30 ROUNDS=2 timeout 600 bash tools/abn.sh "MI_CONV_WS=0 MI_CONV_STREAM=0" "MI_CONV_WS=1 MI_CONV_STREAM=0" "MI_CONV_WS=1 MI_CONV_STREAM=1" "MI_CONV_WS=1 MI_CONV_STREAM=1 MI_C1S_PERCU=1" 2>&1 | tee gpurun_out/c10_ab.log
