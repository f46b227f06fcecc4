#!/bin/bash
O=gpurun_out/blockfn; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_sparseinst.py tests/test_gpu_detr_graph.py tests/test_gpu_detr_meta.py tests/test_gpu_detr.py -q -m gpu > $O/tests2.log 2>&1
tail -8 $O/tests2.log
