"""Which torch ops still launch device work in a training step (developer tool): torch.profiler over a few bench-style steps, printing
every device kernel / memcpy that is not an mmnn:: kernel together with the CPU op that issued it.
    python tools/step_ops.py [size] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mmnn_sts_amd.losses.GradientBlender import GradientBlender  # noqa: E402
from mmnn_sts_amd.losses.losses import CoxPH  # noqa: E402
from mmnn_sts_amd.optim import FusedSGD  # noqa: E402
from mmnn_sts_amd.utils.utils import surv_criterion  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
model = bench.build_model(dev).train()
opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=steps + 8)
blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
inputs, events, durations = bench.synth_batch(dev, 0, 2, size)


def step():
    loss, _ = blender.computeLoss(model(inputs), events, durations)
    loss.backward()
    opt.step()
    sched.step()
    opt.zero_grad()


for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
rows = {}
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU or "mmnn::" in e.name:
        continue
    # innermost CPU op whose interval contains the launch (correlated through time on the CPU side is not exposed: use the kernel's
    # linked CPU parent when the profiler provides one)
    parent = getattr(e, "cpu_parent", None)
    chain = []
    while parent is not None and len(chain) < 6:
        chain.append(parent.name)
        parent = parent.cpu_parent
    key = (e.name[:70], " <- ".join(chain[:5]))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print(f"non-mmnn device activity over {steps} steps at {size}^3:")
for (name, chain), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"{n / steps:6.1f}x/step {us / max(n, 1):7.1f} us  {name}\n        {chain}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
