#!/bin/bash
# same-box env A/B of a secondary bench config: CFG=sparseinst|detr tools/ab_cfg.sh "VAR=a" "VAR=b X=1" ...  (interleaved, ROUNDS times)
S=${STEPS:-20}; R=${ROUNDS:-2}; CFG=${CFG:-sparseinst}
run() { env $1 python bench.py --config $CFG --steps $S --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-48s' % '$1', d['value'], d['ms_per_step'])"; }
for i in $(seq $R); do for e in "$@"; do run "$e"; done; done
