"""Single-rank RCCL exercise for tests/test_dp_gpu.py: the "nccl" (= RCCL) process group, the flat 45 MB gradient bucket and
the coalesced tail bucket go through the same calls bench.py --gpus N makes (the 1-GPU box cannot host a second RCCL rank)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _dp_worker as W  # noqa: E402


def main():
    from mmnn_sts_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    mm = W.build_model(dev)
    W.backward_micro_batch(mm, 0, dev)
    before = {k: p.grad.clone() for k, p in mm.named_parameters() if p.grad is not None}
    buckets = D.gradient_buckets(mm)
    assert buckets[0].numel() > 3_000_000 and buckets[0].data_ptr() == mm.image_model.model.backbone.flat_grad.data_ptr()
    D.allreduce_gradients(mm, force=True)              # SUM over one rank = identity, but every RCCL call is really made
    for t in list(mm.parameters())[:3]:
        dist.broadcast(t.data, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    for k, p in mm.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, before[k]), k
    print("RCCL_OK", dist.get_backend(), len(buckets))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
