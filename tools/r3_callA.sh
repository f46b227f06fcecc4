#!/bin/bash
O=gpurun_out/probe; mkdir -p $O
timeout 120 python tools/torch_ops_probe.py detr > $O/ops_detr.txt 2>&1; head -50 $O/ops_detr.txt
timeout 120 python tools/torch_ops_probe.py sparseinst > $O/ops_si.txt 2>&1; head -3 $O/ops_si.txt
CFG=sparseinst bash tools/prof_secondary.sh > $O/prof_si.txt 2>&1; tail -30 $O/prof_si.txt
