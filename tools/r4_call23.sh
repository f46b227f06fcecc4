#!/bin/bash
# deterministic mask statistics: SparseInst tests + run-to-run reproducibility of the captured step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c23; mkdir -p $O
timeout 900 python -m pytest -q -m gpu tests/test_gpu_sparseinst.py > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for r in 1 2 3 4 5 6; do
  timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | val "sparseinst"
done
