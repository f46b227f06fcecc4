#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv3x3_ws.py -x -q 2>&1 | tail -25 > gpurun_out/c3_test_ws.log
cat gpurun_out/c3_test_ws.log | tail -5
STEPS=30 ROUNDS=2 timeout 600 bash tools/abn.sh "MI_CONV_WS=0" "MI_CONV_WS=1" "MI_CONV_WS=1 MI_C1S_PERCU=1" > gpurun_out/c3_ab.log 2>&1
cat gpurun_out/c3_ab.log
TOP=150 MI_CONV_WS=1 timeout 300 python tools/layer_table.py > gpurun_out/c3_layers_ws.log 2>&1
timeout 600 bash tools/gpu_profile.sh r3b > gpurun_out/c3_prof.log 2>&1
timeout 900 bash tools/gpu_pmc_sq.sh r3b > gpurun_out/c3_pmc.log 2>&1
