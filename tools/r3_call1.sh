#!/bin/bash
# round 3, call 1: streaming 1x1 conv - correctness, A/B against the tile kernel, per-layer table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv1x1_stream.py -x -q 2>&1 | tail -15 > gpurun_out/c1_test_stream.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_parity_bench.py -x -q 2>&1 | tail -8 > gpurun_out/c1_test_rest.log
STEPS=30 ROUNDS=2 timeout 600 bash tools/abn.sh "MI_CONV_STREAM=0" "MI_CONV_STREAM=1" "MI_CONV_STREAM=1 MI_C1S_PERCU=1" > gpurun_out/c1_ab.log 2>&1
TOP=150 MI_CONV_STREAM=1 timeout 300 python tools/layer_table.py > gpurun_out/c1_layers_stream.log 2>&1
TOP=150 MI_CONV_STREAM=0 timeout 300 python tools/layer_table.py > gpurun_out/c1_layers_tile.log 2>&1
cat gpurun_out/c1_test_stream.log gpurun_out/c1_test_rest.log gpurun_out/c1_ab.log
