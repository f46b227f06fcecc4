#!/bin/bash
# round 4, first GPU call: round-3 leftovers (never-executed code), full suite, headline bench, in-graph trace
O=gpurun_out/r4_first; mkdir -p $O
bash tools/boxinfo.sh > $O/boxinfo.log 2>&1; tail -4 $O/boxinfo.log
for c in front_augment_gpu_child mapper_gpu_child jpeg_gpu_child detr_mapper_gpu_child; do
  timeout 240 python tests/$c.py > $O/$c.log 2>&1; echo "$c rc=$? $(tail -1 $O/$c.log)"
done
MI_TEST_UNVERIFIED=1 timeout 200 python -m pytest tests/test_gpu_resnet.py -q -m gpu -k epilogue_fusions > $O/epi.log 2>&1; tail -2 $O/epi.log
for f in 1 0 1 0; do MI_RESNET_EPI_FUSE=$f timeout 60 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -c 300; echo; MI_RESNET_EPI_FUSE=$f timeout 60 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -c 300; echo; done
timeout 120 python tools/input_bench.py 16 20 > $O/input_bench.log 2>&1; tail -7 $O/input_bench.log
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 100 python tools/dump_tags.py > gpurun_out/tags.txt 2>/dev/null
timeout 200 bash tools/gpu_trace.sh r4a
timeout 120 python tools/layer_table.py > $O/layer_table.txt 2>&1; head -3 $O/layer_table.txt
