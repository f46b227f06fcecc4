#!/bin/bash
# A/B an environment switch with the same build, interleaved: tools/abenv.sh "VAR=a" "VAR=b" [steps]
S=${3:-40}
run() { env $1 python bench.py --no-cpu-baseline --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['final_losses'])"; }
for i in 1 2; do run "$1"; run "$2"; done
