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
        # the exposed schedule (one bucket, one weight-gradient group, the single-GPU backward) and the automatic choice
        # land on the overlapped schedule's parameters (to the BatchNorm backward's fp64 accumulation order: the exposed
        # plan may take the one-launch form) - identically on both ranks
        assert r["exposed_params_equal"] and r["exposed_finite"] and r["exposed_buckets"] == 1 and r["exposed_wgrad_groups"] == 1, r
        assert r["exposed_mode"] == "exposed" and r["exposed_choice"] is None and r["exposed_vs_overlap_rel"] < 2e-2, r
        assert r["exposed_bnauto_1step_params_equal"] and r["exposed_bnauto_1step_vs_overlap_rel"] < 2e-2, r
        assert r["auto_params_equal"] and r["auto_finite"] and r["auto_mode"] in ("overlap", "exposed"), r
        ch = r["auto_choice"]
        assert ch["mode"] == r["auto_mode"] and set(ch["selected_on_device"]) == {"overlap_backward_ms", "exposed_backward_ms"}
        sel = ch["selected_on_device"]
        assert (sel["overlap_backward_ms"] <= sel["exposed_backward_ms"]) == (ch["mode"] == "overlap")
        assert r["auto_buckets"] == (3 if ch["mode"] == "overlap" else 1) and r["auto_vs_overlap_rel"] < 2e-2, r
    a, b = (json.load(open(o)) for o in outs)
    assert a["auto_choice"] == b["auto_choice"]                 # both ranks decided on the same two numbers


def test_bench_starts_itself_for_n_gpus():
    """`python bench.py --gpus 2` with NO rendezvous in the environment (the form a driver uses) must launch its own two
    ranks (train_det.py:78-87: d2 `launch(main, num_gpus)`) and print one JSON line from rank 0 - here in the rehearsal
    mode (two gloo ranks on cuda:0), which exercises the same `self_launch` -> torch.distributed.run -> `ddp` block path
    that `--gpus 8` takes on a full node over RCCL."""
    env = dict(os.environ, MI_DIST_SHARE_DEVICE="1", MI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MI_BN_FUSED"):
        env.pop(k, None)
    # (a small batch: two ranks share one device here and the gradient exchange goes through gloo's host staging - the
    # schedule selection alone runs ~20 backward passes with their all-reduces)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "4", "--size", "320", "--no-cpu-baseline", "--no-h2d"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ddp"]["ranks"] == 2 and d["ddp"]["rccl_ranks"] == 0
    sch = d["ddp"]["schedule"]
    assert sch["mode"] in ("overlap", "exposed") and len(d["ddp"]["buckets_MB"]) == (3 if sch["mode"] == "overlap" else 1)
    assert sch["selected_on_device"]["overlap_backward_ms"] > 0 and sch["selected_on_device"]["exposed_backward_ms"] > 0
    assert all(x == x for x in d["config"]["final_losses"])
