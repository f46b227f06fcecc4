#!/bin/bash
# round-4 evidence: full GPU suite + smoke + default bench (+ its rocprofv3 kernel stats and counter passes) + the two secondary benches
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json
cd /tmp
MI_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_prof.log 2>&1
find $O/raw -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/raw
head -8 $O/kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --config detr 2>/dev/null | tail -1 > $O/bench_detr.json; cut -c1-300 $O/bench_detr.json
timeout 300 python bench.py --config sparseinst 2>/dev/null | tail -1 > $O/bench_sparseinst.json; cut -c1-300 $O/bench_sparseinst.json
bash tools/gpu_pmc.sh r04f > $O/pmc_log.txt 2>&1; tail -6 $O/pmc_log.txt | cut -c1-200
cp gpurun_out/pmc_r04f/hbm_traffic.csv $O/hbm_traffic_pmc.csv 2>/dev/null
rocm-smi --showclocks 2>/dev/null | head -12 > $O/box.txt; git rev-parse HEAD >> $O/box.txt 2>/dev/null
