"""GPU (-m gpu): the device side of the data-parallel step, rehearsed on ONE GPU - two gloo ranks share cuda:0 (RCCL
cannot run on this pool's 1-GPU boxes; train_det.py:73-87 -> d2 create_ddp_model is what it replaces).  Asserted on the
device: graphs on, three captured backward segments with a bucket all-reduce after each, reduced gradient == sum of the two
ranks' single-list gradients, identical parameters after real updates, and the one-launch BatchNorm backward resolving to
the two-pass form under world > 1 unless forced."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("MI_BN_FUSED", None)
        out = str(tmp_path / f"rank{r}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_rehearsal_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    for out in outs:
        r = json.load(open(out))
        assert r["single_segments"] <= 2                       # world 1: one backward list (+ a tail without gradients)
        assert len(r["segments"]) >= 3 and sum(1 for s in r["segments"] if s[2] is not None) == 3, r["segments"]
        assert r["wgrad_groups"] == 3 and r["graphs"] >= 3     # three staged weight-gradient groups, >= 3 captured segments
        assert r["bn_fused_under_ddp"] is False                # two-pass unless MI_BN_FUSED forces the grid-barrier kernel
        # reduced gradient == sum of the ranks' local gradients (same kernels, same inputs: fp32 summation order of the
        # host-staged gloo reduction only)
        assert r["reduced_vs_sum_rel"] < 1e-5 and r["reduced_vs_sum_max"] < 1e-5, r
        assert r["sum_norm"] > r["local_norm"] * 0.5
        assert r["params_equal"] and r["finite"] and r["params_moved"] > 1e-6, r
