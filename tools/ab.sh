#!/bin/bash
# A/B two builds of libmi355det in ONE gpurun call (box-to-box variance is ~4 %): tools/ab.sh libA.so libB.so [steps]
A=${1:-yolov7_d2_amd/libA.so}; B=${2:-yolov7_d2_amd/libmi355det.so}; S=${3:-40}
for i in 1 2; do for v in $A $B; do
  MI355_LIB=$PWD/$v python bench.py --no-cpu-baseline --steps $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done
