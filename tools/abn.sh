#!/bin/bash
# A/B/C... environment switches with the same build, interleaved, in ONE gpurun call: tools/abn.sh "VAR=a" "VAR=b X=1" ...
# (box-to-box spread of the pool is several percent: only numbers from one call are comparable)
S=${STEPS:-40}; R=${ROUNDS:-2}
run() { env MI_BENCH_NO_PMC=1 $1 python bench.py --no-cpu-baseline --no-h2d --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s' % '$1', d['value'], d['ms_per_step'], 'roofline', r['kernel'][:28], r['avg_launch_ms'], d['config']['final_losses'])"; }
for i in $(seq $R); do for e in "$@"; do run "$e"; done; done
