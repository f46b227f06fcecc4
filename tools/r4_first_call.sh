#!/bin/bash
# first GPU call of the next round: everything that was written or changed after round 3's GPU minutes were spent, then the
# usual verification.  ~6-8 GPU minutes.  usage: gpurun --timeout 900 -- 'bash tools/r4_first_call.sh'
O=gpurun_out/r4_first; mkdir -p $O
for c in front_augment_gpu_child mapper_gpu_child jpeg_gpu_child detr_mapper_gpu_child; do
  timeout 240 python tests/$c.py > $O/$c.log 2>&1; echo "$c rc=$? $(tail -1 $O/$c.log)"
done
timeout 120 python tools/input_bench.py 16 20 > $O/input_bench.log 2>&1; tail -7 $O/input_bench.log
timeout 600 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
