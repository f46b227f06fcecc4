#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
STEPS=30 ROUNDS=2 timeout 600 bash tools/abn.sh "MI_CONV_STREAM=1" "MI_CONV_STREAM=1 MI_DEBUG_NOATOM=1" "MI_CONV_STREAM=1 MI_C1S_PERCU=1" "MI_CONV_STREAM=1 MI_C1S_PERCU=1 MI_DEBUG_NOATOM=1" "MI_CONV_STREAM=0 MI_DEBUG_NOATOM=1" > gpurun_out/c2_ab.log 2>&1
MI_CONV_STREAM=1 timeout 600 python -m pytest "tests/test_gpu_step.py::test_grouped_launches_equal_separate_launches" -x -q 2>&1 | tail -40 > gpurun_out/c2_test_s1.log
MI_CONV_STREAM=0 timeout 600 python -m pytest "tests/test_gpu_step.py::test_grouped_launches_equal_separate_launches" -x -q 2>&1 | tail -40 > gpurun_out/c2_test_s0.log
MI_C1S_PERCU=1 timeout 600 bash tools/gpu_profile.sh r3a > gpurun_out/c2_prof.log 2>&1
cat gpurun_out/c2_ab.log; tail -5 gpurun_out/c2_test_s1.log gpurun_out/c2_test_s0.log
