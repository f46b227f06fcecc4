#!/bin/bash
# census of the captured DETR / SparseInst steps: per-dispatch trace of replays + call sites of torch's own kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_trace.sh detr --config detr --steps 8 --warmup 3
bash tools/gpu_trace.sh si --config sparseinst --steps 8 --warmup 3
timeout 300 python tools/torch_ops_probe.py detr > gpurun_out/torch_ops_detr.txt 2>&1; tail -3 gpurun_out/torch_ops_detr.txt
timeout 300 python tools/torch_ops_probe.py sparseinst > gpurun_out/torch_ops_si.txt 2>&1; tail -3 gpurun_out/torch_ops_si.txt
