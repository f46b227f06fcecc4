#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ref_nms.py tests/test_gpu_ddp_rehearsal.py tests/test_gpu_resnet.py tests/test_gpu_sparseinst.py tests/test_gpu_detr.py "tests/test_gpu_step.py::test_backward_keeps_autograd_accumulation_semantics" -q -m gpu 2>&1 | tail -40 > gpurun_out/c12_tests.log
tail -40 gpurun_out/c12_tests.log
