#!/bin/bash
# interleaved A/B/C... of environment settings with one build: tools/abenvn.sh STEPS "VAR=a" "VAR=b X=1" ...
S=$1; shift
run() { env $1 MI_BN_FUSED=${MI_BN_FUSED:-1} python bench.py --no-cpu-baseline --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['final_losses'])"; }
for i in 1 2; do for v in "$@"; do run "$v"; done; done
