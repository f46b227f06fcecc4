#!/bin/bash
# in-projection as one node + one pack launch per captured step + row_scale in the wgrad reduction + dropout-add: tests, A/B, sites
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_c16; mkdir -p $O
timeout 900 python -m pytest -q -m gpu tests > $O/tests.log 2>&1; tail -15 $O/tests.log | cut -c1-300
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"; }
for f in 1 0 1 0; do
  MI_BATCH_PACK=$f timeout 120 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>$O/detr_err.log | val "detr BATCH_PACK=$f"
  MI_BATCH_PACK=$f timeout 120 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>$O/si_err.log | val "sparseinst BATCH_PACK=$f"
done
timeout 300 python tools/py_kernel_sites.py detr > $O/sites_detr.txt 2>&1; tail -3 $O/sites_detr.txt
timeout 300 python tools/py_kernel_sites.py sparseinst > $O/sites_si.txt 2>&1; tail -3 $O/sites_si.txt
