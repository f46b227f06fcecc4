#!/bin/bash
# same-box A/B of the weight-gradient side queue (runs on the GPU box): default | in-graph async branch | side queue
# (separate graphs) unmasked / masked 32, 64, 96 CUs, slices of every XCD and whole XCDs, with and without the complement
# mask on the chain's stream
S=${STEPS:-40}
run() { env MI_BENCH_NO_PMC=1 $1 python bench.py --no-cpu-baseline --no-h2d --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-64s' % '$1', d['value'], d['ms_per_step'], d['config']['final_losses'])"; }
for r in 1 2; do
run "MI_X=default"


run "MI_WGRAD_SIDE=2"
run "MI_WGRAD_SIDE=3"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=32"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=64"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=96"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=64x"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=64 MI_MAIN_CUMASK=1"
run "MI_WGRAD_SIDE=3 MI_WGRAD_CUMASK=64x MI_MAIN_CUMASK=1"
run "MI_WGRAD_SIDE=2 MI_WGRAD_CUMASK=64"
run "MI_WGRAD_SIDE=4 MI_WGRAD_CUMASK=64"
run "MI_WGRAD_SIDE=4 MI_WGRAD_CUMASK=96 MI_MAIN_CUMASK=1"
done
