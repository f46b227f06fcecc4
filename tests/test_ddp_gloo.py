"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (bucket planning from the backward command
list, bucketed all-reduce, parameter broadcast).  One process per rank, rendezvous on 127.0.0.1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolov7_d2_amd.parallel import GradReducer, broadcast_params, plan_buckets


def test_plan_buckets_orders_by_completion():
    # 3 commands; cmd0 writes the tail of the arena (head grads), cmd2 writes the front (stem grads)
    writes = [[(800 * 4, 1000 * 4)], [(300 * 4, 800 * 4)], [(0, 300 * 4)]]
    b = plan_buckets(1000, writes, 3)
    assert [x[2] for x in b] == sorted(x[2] for x in b)
    last = {(lo, hi): k for lo, hi, k in b}
    tail = [k for (lo, hi), k in last.items() if hi == 1000][0]
    front = [k for (lo, hi), k in last.items() if lo == 0][0]
    assert tail < front and b[-1][0] == 0              # head gradients complete first, the stem's bucket last
    assert sorted((x[0], x[1]) for x in b)[0][0] == 0 and sorted((x[0], x[1]) for x in b)[-1][1] == 1000
    red = GradReducer(torch.zeros(1000), b)
    segs = red.segments(3)
    assert segs[0][0] == 0 and segs[-1][1] == 3
    assert all(s[1] >= s[0] for s in segs)
    covered = sorted((s[2] for s in segs if s[2]), key=lambda t: t[0])
    assert covered[0][0] == 0 and covered[-1][1] == 1000


def test_plan_buckets_with_explicit_bounds():
    """the split weight-gradient groups cut the arena where the early (neck + head) group starts: that bucket completes
    at the early group's command, the backbone bucket at the end"""
    writes = [[(600 * 4, 1000 * 4)], [], [(0, 600 * 4)]]          # cmd0 = early wgrad group, cmd2 = late group
    b = plan_buckets(1000, writes, 3, bounds=[600])
    assert b == [(600, 1000, 0), (0, 600, 2)]
    red = GradReducer(torch.zeros(1000), b)
    assert red.segments(3) == [(0, 1, (600, 1000)), (1, 3, (0, 600))]
    assert plan_buckets(1000, writes, 2, bounds=[0, 1000, 5000]) == plan_buckets(1000, writes, 1)   # degenerate bounds


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1003
    params = torch.full((n,), float(rank + 1))
    broadcast_params(params)
    assert float(params[0]) == 1.0 and float(params[-1]) == 1.0     # rank 0's values everywhere
    grad = torch.arange(n, dtype=torch.float32) * (rank + 1)
    writes = [[(600 * 4, n * 4)], [(0, 600 * 4)]]
    red = GradReducer(grad, plan_buckets(n, writes, 2))
    for (lo, hi, bucket) in red.segments(2):
        red.reduce_bucket(bucket)
    red.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(grad, expect)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)
