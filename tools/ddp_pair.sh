#!/bin/bash
# two gloo ranks of tests/detr_ddp_worker.py on cuda:0; prints their JSON verdicts.  usage: tools/ddp_pair.sh <tag> [ENV=VAL ...]
tag=$1; shift
mkdir -p gpurun_out
port=$((20000 + RANDOM % 20000))
pids=()
for r in 0 1; do
  env MI_TEST_STACKS=${DDP_STACKS:-100} "$@" RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout ${DDP_TIMEOUT:-150} python tests/detr_ddp_worker.py gpurun_out/ddp_${tag}_rank$r.json > gpurun_out/ddp_${tag}_rank$r.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for r in 0 1; do echo "== $tag rank $r"; cat gpurun_out/ddp_${tag}_rank$r.json 2>/dev/null || tail -20 gpurun_out/ddp_${tag}_rank$r.log; echo; done
