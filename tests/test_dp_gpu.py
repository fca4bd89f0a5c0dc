"""Data parallelism over patients on the real model: W ranks x one micro-batch each followed by
all-reduce(SUM) must equal the reference's own accumulation of the same W micro-batches on one rank (main.py:403-407,469,478-481)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_allreduce_equals_sequential_accumulation(tmp_path):
    from tests import _dp_worker as W
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["world"] == r1["world"] == 2
    # overlapped reduction (one all-reduce per dense block, issued from inside the backward): same bits as reducing afterwards, for a
    # single backward and for a two-micro-batch accumulation window whose last backward is armed
    for r in (r0, r1):
        assert r["overlap_equal"] and r["window_equal"], (r["overlap_equal"], r["window_equal"])
        assert r["early_works"] == 3 and len(r["ranges"]) == 3                       # TinyDensenet: three dense blocks
        assert r["ranges"][0][1] == r["flat"] and r["ranges"][-1][0] == 0            # last block (+ norm5) first, stem with block 1
        assert all(r["ranges"][i][0] == r["ranges"][i + 1][1] for i in range(2))     # the ranges tile the flat buffer, downwards
    assert r0["loss"] != r1["loss"]                       # the ranks really worked on different patients
    # one rank, the same two micro-batches accumulated (no zero_grad in between)
    dev = torch.device("cuda", 0)
    mm = W.build_model(dev)
    l0 = W.backward_micro_batch(mm, 0, dev)
    l1 = W.backward_micro_batch(mm, 1, dev)
    assert abs(l0 - r0["loss"]) < 1e-6 * abs(l0) and abs(l1 - r1["loss"]) < 1e-6 * abs(l1)
    seq = {k: p.grad.detach().cpu() for k, p in mm.named_parameters() if p.grad is not None}
    assert set(seq) == set(r0["grads"]) == set(r1["grads"]) and len(seq) == 175     # TinyDensenet fusion: 179 tensors - 4 without grad
    gl2 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in seq.values())))
    for k, g in seq.items():
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k                     # both ranks hold the same reduced gradient
        err = float((r0["grads"][k].double() - g.double()).norm())
        assert err <= 2e-6 * float(g.double().norm()) + 1e-7 * gl2, (k, err, float(g.norm()))


def test_rccl_single_rank_buckets():
    """The RCCL ("nccl") backend itself: process group, flat 45 MB-style bucket, coalesced tail bucket, broadcast, barrier."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK nccl" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
