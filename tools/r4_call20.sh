#!/bin/bash
# post-norm residual as one node + parallel LN parameter reduction + MIN_TILES 2: DETR tests, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c20; mkdir -p $O
timeout 900 python -m pytest -q -m gpu tests/test_gpu_detr.py tests/test_gpu_detr_meta.py tests/test_gpu_detr_graph.py > $O/tests.log 2>&1; tail -8 $O/tests.log | cut -c1-300
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for f in 1 2; do
  timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>$O/detr_err.log | val "detr"
done
timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>$O/si_err.log | val "sparseinst"
