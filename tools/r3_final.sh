#!/bin/bash
# full GPU suite + smoke + default bench + rocprofv3 kernel stats of the bench (round-3 evidence)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r03f/pytest_gpu.txt; cat gpurun_out/r03f/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r03f/bench.json; cut -c1-400 gpurun_out/r03f/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03f/raw -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03f/bench_prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r03f/raw -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03f/kernel_stats.csv \;
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03f/raw
head -12 $GRAFT_REPO_ROOT/gpurun_out/r03f/kernel_stats.csv | cut -c1-160
