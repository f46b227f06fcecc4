#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_bn_fused.py -q 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-400 | head -30
