#!/bin/bash
# HEAD build (libA) vs working build, interleaved in ONE gpurun call
S=${1:-40}
run() { MI355_LIB=$PWD/$1 python bench.py --no-cpu-baseline --steps $S --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do run yolov7_d2_amd/libA.so; run yolov7_d2_amd/libmi355det.so; done
