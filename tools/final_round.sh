#!/bin/bash
# on the GPU box: the round's evidence in one call -> gpurun_out/final_<tag>/   usage: tools/final_round.sh <tag> [skip-tests]
set -u
TAG=${1:-r06}; SKIP=${2:-}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP" ]; then timeout 1500 python -m pytest tests/ -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt; fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json
timeout 200 python bench.py --config detr 2>/dev/null | tail -1 > $O/bench_detr.json; cut -c1-160 $O/bench_detr.json
timeout 200 python bench.py --config sparseinst 2>/dev/null | tail -1 > $O/bench_sparseinst.json; cut -c1-160 $O/bench_sparseinst.json
bash tools/gpu_profile.sh $TAG > /dev/null 2>&1; cp gpurun_out/prof_$TAG/kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
bash tools/gpu_pmc.sh $TAG > /dev/null 2>&1; cp gpurun_out/pmc_$TAG/hbm_traffic.csv $O/hbm_traffic_pmc.csv 2>/dev/null
ls -la $O
