#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_conv_bn_fused.py tests/test_gpu_conv1x1_stream.py tests/test_gpu_conv3x3_ws.py tests/test_abi_and_config.py -q 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -30
