#!/bin/bash
O=gpurun_out/r4_c9; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -6 $O/tests.log | cut -c1-300
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('value_incl_h2d'), d['config']['bn_backward'], d['roofline']['kernel'], d['roofline']['frac'])"
for v in "MI_CONV_BN_FUSE=0" "MI_CONV_BN_FUSE=1 MI_CONV_BN_FUSE_MAXPIX=1600" "MI_CONV_BN_FUSE=1 MI_CONV_BN_FUSE_MAXPIX=400" "MI_CONV_BN_FUSE=0" "MI_CONV_BN_FUSE=1 MI_CONV_BN_FUSE_MAXPIX=1600" "MI_CONV_BN_FUSE=1 MI_CONV_BN_FUSE_MAXPIX=400"; do
  env $v python bench.py --no-cpu-baseline --no-h2d --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['config']['final_losses'])"
done
