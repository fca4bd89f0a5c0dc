"""Data parallelism over patients: one process per GPU, SUM all-reduce of the gradients at the accumulation boundary
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" in the CPU tests).

The reference sums UN-normalised micro-batch gradients until 64 patients were seen (main.py:403,469,478-481); W ranks x
(32/W) micro-batches of 2 followed by all-reduce(SUM) is that same update.  No other collective exists on
the path: BN statistics and the Cox risk set are per micro-batch, hence per rank.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl"):
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; initialises the process group when world > 1."""
    import os
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gradient_buckets(model: torch.nn.Module) -> List[torch.Tensor]:
    """The tensors to reduce: each native backbone's flat gradient buffer as ONE bucket (45 MB for DenseNet121), and every
    other parameter gradient (the ~30 tail tensors) individually."""
    buckets, seen = [], set()
    for m in model.modules():
        fg = getattr(m, "flat_grad", None)
        if isinstance(fg, torch.Tensor) and hasattr(m, "flat_parameters"):
            buckets.append(fg)
            seen.update(id(p) for p in m.parameters())
    for p in model.parameters():
        if id(p) not in seen and p.grad is not None:
            buckets.append(p.grad)
    return buckets


def allreduce_gradients(model: torch.nn.Module, group=None, force: bool = False) -> None:
    """SUM all-reduce of every gradient; the small tensors travel as one coalesced flat buffer.  A single-rank group is a
    no-op unless `force` (used by the tests to push the buckets through RCCL on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    buckets = gradient_buckets(model)
    big = [b for b in buckets if b.numel() >= 1 << 16]
    small = [b for b in buckets if b.numel() < 1 << 16]
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in big]
    if small:
        flat = torch.cat([b.reshape(-1) for b in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for b in small:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for w in works:
        w.wait()


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical initial weights / buffers on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
