#!/bin/bash
CFG=detr bash tools/prof_secondary.sh > gpurun_out/prof_detr_r03h.txt 2>&1; tail -28 gpurun_out/prof_detr_r03h.txt | head -3
CFG=sparseinst bash tools/prof_secondary.sh > gpurun_out/prof_si_r03h.txt 2>&1; tail -28 gpurun_out/prof_si_r03h.txt | head -3
