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


def _small_bucket(small: List[torch.Tensor]):
    """(flat buffer, [(refs, n)]) for the coalesced bucket of the small gradients: gathered and scattered back by ONE HIP launch each
    (`mmnn_multi_copy`, pointer table as kernel argument) when they live on the GPU; torch.cat / copy_ on the CPU (gloo tests)."""
    total, offs = 0, []
    for b in small:
        offs.append(total)
        total += b.numel()
    flat = torch.empty((total,), device=small[0].device, dtype=small[0].dtype)
    return flat, offs


def _multi_copy(small, flat, offs, scatter: bool) -> None:
    from . import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for lo in range(0, len(small), _lib.MULTI_MAX):
        chunk = small[lo:lo + _lib.MULTI_MAX]
        refs = (_lib.TensorRef * len(chunk))()
        for r, b, o in zip(refs, chunk, offs[lo:lo + _lib.MULTI_MAX]):
            r.param, r.grad, r.count, r.flat_offset, r.first_step = None, b.data_ptr(), b.numel(), o, 0
        _lib.check(L.mmnn_multi_copy(refs, len(chunk), flat.data_ptr(), int(scatter), st), "multi_copy")


def _reduce_small(small: List[torch.Tensor], group) -> None:
    if not small:
        return
    native = all(b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() for b in small)
    flat, offs = _small_bucket(small)
    if native:
        _multi_copy(small, flat, offs, scatter=False)
    else:
        torch.cat([b.reshape(-1) for b in small], out=flat)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if native:
        _multi_copy(small, flat, offs, scatter=True)
    else:
        for b, o in zip(small, offs):
            b.copy_(flat[o:o + b.numel()].view_as(b))


def allreduce_gradients(model: torch.nn.Module, group=None, force: bool = False) -> None:
    """SUM all-reduce of every gradient after the backward; the small tensors travel as one coalesced flat buffer.  A single-rank
    group is a no-op unless `force` (used by the tests to push the buckets through RCCL on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    buckets = gradient_buckets(model)
    big = [b for b in buckets if b.numel() >= 1 << 16]
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in big]
    _reduce_small([b for b in buckets if b.numel() < 1 << 16], group)
    for w in works:
        w.wait()


class OverlappedGradientReducer:
    """The same reduction, started early: each native backbone reports, from inside its backward, the ranges of its flat gradient
    buffer that are final (dense block 4 + norm5, then block 3 + transition 3, ... -- `_Backbone.set_grad_ready_hook`), and the
    SUM all-reduce of a range is issued at once (asynchronously: RCCL runs it on its own stream behind an event of the compute
    stream) while the kernels of the next block execute.  ~90 % of DenseNet121's parameters sit in blocks 2-4, whose gradients are
    final ~3.5 ms before the backward ends.

        reducer = OverlappedGradientReducer(model)
        ...
        reducer.arm()            # the NEXT backward completes an accumulation window (main.py:478: every 64 / batch micro-batches)
        loss.backward()
        reducer.finish()         # reduces whatever was not reduced early (the small tensors), waits for everything

    Un-armed backwards (the other micro-batches of the window) only accumulate locally.  Results are bit-identical to
    `allreduce_gradients` after the backward: the same buffers are summed over the same ranks, only earlier and in pieces
    (an element-wise SUM does not depend on how the buffer is cut)."""

    def __init__(self, model: torch.nn.Module, group=None, force: bool = False):
        self.model, self.group, self.force = model, group, force
        self._armed = False
        self._works = []
        self._done = {}            # id(backbone) -> [(begin, end)] reduced early in this window
        self.backbones = [m for m in model.modules() if hasattr(m, "set_grad_ready_hook")]
        for bb in self.backbones:
            bb.set_grad_ready_hook(self._on_range)

    def _active(self) -> bool:
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def arm(self) -> None:
        self._armed = True

    def _on_range(self, bb, begin: int, end: int) -> None:
        if not (self._armed and self._active()) or end <= begin:
            return
        self._works.append(dist.all_reduce(bb.flat_grad[begin:end], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._done.setdefault(id(bb), []).append((begin, end))

    def finish(self) -> None:
        if self._active():
            seen = set()
            for bb in self.backbones:
                seen.update(id(p) for p in bb.parameters())
                fg = bb.flat_grad
                if fg is None:
                    continue
                covered = sorted(self._done.get(id(bb), []))
                pos = 0
                for b, e in covered + [(fg.numel(), fg.numel())]:      # whatever the hooks did not cover (un-armed backward: everything)
                    if b > pos:
                        self._works.append(dist.all_reduce(fg[pos:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    pos = max(pos, e)
            _reduce_small([p.grad for p in self.model.parameters() if id(p) not in seen and p.grad is not None], self.group)
            for w in self._works:
                w.wait()
        self._works, self._done, self._armed = [], {}, False

    def detach(self) -> None:
        for bb in self.backbones:
            bb.set_grad_ready_hook(None)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical initial weights / buffers on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
