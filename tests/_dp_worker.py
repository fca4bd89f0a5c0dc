"""One data-parallel rank of tests/test_dp_gpu.py: real MultiModalModel on the MI355X, gloo process group (two ranks share the one
card of the GPU box; on an 8-GPU node the same code runs over RCCL).  Usage: RANK/WORLD_SIZE/MASTER_* in the env, argv[1] = out dir."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import restatement as R  # noqa: E402
from tests._util import N_CLIN, synth_sd  # noqa: E402
from oracle import synth  # noqa: E402

S = 32


def build_model(dev):
    from mmnn_sts_amd.models.densenet import TinyDensenet
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    cfg = R.DenseNetCfg(in_channels=2, block_config=(6, 12, 4))
    img = TinyDensenet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=0.0)
    mm = MultiModalModel(img, [f"p{i}" for i in range(N_CLIN)], 2, 12, blend=True)
    mm.load_state_dict(synth_sd(R.multimodal_schema(cfg, N_CLIN, 2, 12), "dp."), strict=True)
    for m in mm.modules():
        if m.__class__.__name__.startswith("Dropout"):
            m.p = 0.0
    return mm.to(dev).train()


def micro_batch(i, dev):
    image = torch.from_numpy(synth.uniform(f"dp/image/{i}", (2, 2, S, S, S))).to(dev)
    clinical = torch.from_numpy(synth.uniform(f"dp/clin/{i}", (2, N_CLIN))).to(dev)
    ev = torch.tensor([[1, 0], [1, 1]], device=dev) if i % 2 else torch.tensor([[1, 1], [0, 1]], device=dev)
    du = torch.tensor([[100 + 7 * i, 250], [300, 50 + 3 * i]], device=dev)
    return {"image": image, "clinical": clinical}, ev, du


def backward_micro_batch(mm, i, dev):
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    from mmnn_sts_amd.losses.losses import CoxPH
    from mmnn_sts_amd.utils.utils import surv_criterion
    x, ev, du = micro_batch(i, dev)
    gb = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
    loss, _ = gb.computeLoss(mm(x), ev, du)
    loss.backward()
    return float(loss.detach())


def main():
    from mmnn_sts_amd import distributed as D
    out = sys.argv[1]
    rank, world, _ = D.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    mm = build_model(dev)
    D.broadcast_parameters(mm)
    loss = backward_micro_batch(mm, rank, dev)
    D.allreduce_gradients(mm)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in mm.named_parameters() if p.grad is not None}
    # The same window again with the all-reduce OVERLAPPED with the backward (per dense block, from inside backward()): bit-identical.
    # Second pass: a two-micro-batch accumulation window where only the LAST backward is armed, against reducing after the window.
    overlap_equal, ranges = True, []
    mm.zero_grad(set_to_none=True)
    bb = mm.image_model.model.backbone
    bb.mark_grads_stale()
    red = D.OverlappedGradientReducer(mm)
    seen = red._on_range
    def spy(b, begin, end):
        ranges.append((begin, end)); seen(b, begin, end)
    bb.set_grad_ready_hook(spy)
    red.arm()
    backward_micro_batch(mm, rank, dev)
    early = len(red._works)
    ranges = list(ranges)                                         # (the spy stays installed for the window passes below)
    first_pass_ranges = list(ranges)
    red.finish()
    torch.cuda.synchronize()
    for k, p in mm.named_parameters():
        if p.grad is not None and not torch.equal(p.grad.detach().cpu(), grads[k]):
            overlap_equal = False
    window = {}
    for mode in ("after", "overlap"):
        mm.zero_grad(set_to_none=True)
        bb.mark_grads_stale()
        backward_micro_batch(mm, 2 * rank, dev)                   # accumulates locally: the reducer is not armed
        if mode == "overlap":
            red.arm()
        backward_micro_batch(mm, 2 * rank + 1, dev)
        red.finish()                                              # un-armed: reduces everything here
        torch.cuda.synchronize()
        window[mode] = {k: p.grad.detach().cpu().clone() for k, p in mm.named_parameters() if p.grad is not None}
    window_equal = all(torch.equal(window["after"][k], window["overlap"][k]) for k in window["after"])
    torch.save({"loss": loss, "grads": grads, "world": torch.distributed.get_world_size(), "overlap_equal": overlap_equal, "ranges": first_pass_ranges,
                "early_works": early, "window_equal": window_equal, "flat": bb.flat_grad.numel()}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
