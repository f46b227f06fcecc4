#!/bin/bash
A=${1:-yolov7_d2_amd/libA.so}; B=${2:-yolov7_d2_amd/libmi355det.so}; S=${3:-40}
run() { MI355_LIB=$PWD/$1 MI_CONV_TUNE=$2 python bench.py --no-cpu-baseline --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 tune=$2', d['value'], d['ms_per_step'])"; }
for i in 1 2; do run $A 0; run $B 0; run $B 1; done
