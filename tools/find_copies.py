"""Which Python lines cause device-to-device copies in a training step (developer tool)."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmnn_sts_amd.losses.GradientBlender import GradientBlender
from mmnn_sts_amd.losses.losses import CoxPH
from mmnn_sts_amd.optim import FusedSGD
from mmnn_sts_amd.utils.utils import surv_criterion

dev = torch.device("cuda", 0)
model = bench.build_model(dev).train()
opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
inputs, events, durations = bench.synth_batch(dev, 0)
def step():
    out = model(inputs)
    loss, _ = blender.computeLoss(out, events, durations)
    loss.backward()
    opt.step(); opt.zero_grad()
for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    n = e.name
    if "copy_" in n or "Memcpy" in n or "memcpy" in n or "clone" in n or "fill_" in n or "zero_" in n:
        st = [s for s in (e.stack or []) if "mmnn_sts_amd" in s or "bench.py" in s or "find_copies" in s]
        c[(n, st[0] if st else "?")] += 1
for (n, s), k in c.most_common(40):
    print(k, n, s)
