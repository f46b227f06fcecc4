#!/bin/bash
S=${1:-40}
run() { env $2 MI355_LIB=$PWD/$1 MI_CONV_TUNE=0 python bench.py --no-cpu-baseline --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['ms_per_step'], d['config']['sum_kernel_ms_per_step'])"; }
for i in 1 2; do
run yolov7_d2_amd/libA.so X=1
run yolov7_d2_amd/libmi355det.so X=1
run yolov7_d2_amd/libmi355det.so MI_CONV_MAXKC=64
run yolov7_d2_amd/libmi355det.so MI_CONV_OLDCFG=1
done
