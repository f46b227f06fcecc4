#!/bin/bash
# on the GPU box: the collective-footprint probe for both BatchNorm-backward forms -> gpurun_out/ddp_footprint.txt
mkdir -p gpurun_out
for m in 1 0; do MI_BN_FUSED=$m timeout 150 python tools/ddp_footprint_probe.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/ddp_footprint.txt
