#!/bin/bash
# first GPU call of the next round: everything that was written or changed after round 3's GPU minutes were spent, then the
# usual verification.  ~6-8 GPU minutes.  usage: gpurun --timeout 900 -- 'bash tools/r4_first_call.sh'
O=gpurun_out/r4_first; mkdir -p $O
for c in front_augment_gpu_child mapper_gpu_child jpeg_gpu_child detr_mapper_gpu_child; do
  timeout 240 python tests/$c.py > $O/$c.log 2>&1; echo "$c rc=$? $(tail -1 $O/$c.log)"
done
MI_TEST_UNVERIFIED=1 timeout 200 python -m pytest tests/test_gpu_resnet.py -q -m gpu -k epilogue_fusions > $O/epi.log 2>&1; tail -2 $O/epi.log
for f in 1 0; do MI_RESNET_EPI_FUSE=$f timeout 60 python bench.py --config detr --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -c 300; MI_RESNET_EPI_FUSE=$f timeout 60 python bench.py --config sparseinst --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -c 300; done
timeout 120 python tools/input_bench.py 16 20 > $O/input_bench.log 2>&1; tail -7 $O/input_bench.log
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
