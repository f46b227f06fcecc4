#!/bin/bash
# GPU box: in-graph kernel traces of the YOLOX-s step under several environment settings, one gpurun call:
#   tools/ab_trace.sh name1 "ENV=a" name2 "ENV=b X=1" ...   ->  gpurun_out/trace_<name>.csv + gpurun_out/tags_<name>.txt
# locally: python tools/trace_steps.py <name>   (reads tags_<name>.txt when it exists)
while [ $# -ge 2 ]; do
  n=$1; e=$2; shift 2
  env $e python tools/dump_tags.py > gpurun_out/tags_$n.txt 2>/dev/null
  env $e MI_BENCH_NO_PMC=1 tools/gpu_trace.sh $n --no-h2d
done
