#!/bin/bash
# BatchNorm forward: first-trip loads hoisted above the statistics prologue - parity + same-box A/B against the previous build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c18; mkdir -p $O
timeout 900 python -m pytest -q -m gpu tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_parity_bench.py tests/test_gpu_conv_bn_fused.py > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-300
STEPS=40 ROUNDS=3 bash tools/abn.sh "MI355_LIB=yolov7_d2_amd/libmi355det_prev.so" "MI355_LIB=yolov7_d2_amd/libmi355det.so"
