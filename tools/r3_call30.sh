#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity_bench.py tests/test_gpu_widths.py -x -q 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -20
for v in 0 1 0 1; do
  MI_CSP_DGRAD_PAIR=$v MI_BENCH_LIVE_PMC=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('PAIR=$v', d['ms_per_step'], d['value'], d['roofline']['kernel'][:40], d['roofline']['frac'])"
done
