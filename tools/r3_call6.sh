#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv3x3_ws.py -q 2>&1 | tail -15 > gpurun_out/c6_test_ws.log
tail -6 gpurun_out/c6_test_ws.log
timeout 600 python tools/ws_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c6_probe.log
STEPS=30 ROUNDS=2 timeout 600 bash tools/abn.sh "MI_CONV_WS=0" "MI_CONV_WS=1" > gpurun_out/c6_ab.log 2>&1
cat gpurun_out/c6_ab.log
