#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f; export TMPDIR=/tmp
bash tools/gpu_pmc.sh r03f > gpurun_out/r03f/pmc_log.txt 2>&1; tail -8 gpurun_out/r03f/pmc_log.txt | cut -c1-200
bash tools/gpu_pmc_sq.sh r03f > gpurun_out/r03f/pmcsq_log.txt 2>&1; tail -3 gpurun_out/r03f/pmcsq_log.txt | cut -c1-200
timeout 600 python bench.py --config detr 2>/dev/null | tail -1 > gpurun_out/r03f/bench_detr.json; cut -c1-300 gpurun_out/r03f/bench_detr.json
timeout 600 python bench.py --config sparseinst 2>/dev/null | tail -1 > gpurun_out/r03f/bench_sparseinst.json; cut -c1-300 gpurun_out/r03f/bench_sparseinst.json
bash tools/mha_pmc.sh mha_ > gpurun_out/r03f/mha_pmc.txt 2>&1; tail -6 gpurun_out/r03f/mha_pmc.txt | cut -c1-200
