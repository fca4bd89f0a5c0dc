"""Where a training step's wall time goes on the GPU timeline, without a profiler (developer tool): events recorded on the
current stream around the backbone forward / backward C calls of bench.py's step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmnn_sts_amd.losses.GradientBlender import GradientBlender
from mmnn_sts_amd.losses.losses import CoxPH
from mmnn_sts_amd.optim import FusedSGD
from mmnn_sts_amd.utils.utils import surv_criterion

dev = torch.device("cuda", 0)
model = bench.build_model(dev).train()
opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=1000)
blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
inputs, events, durations = bench.synth_batch(dev, 0)
bb = model.image_model.model.backbone
marks = []
of, ob = bb._run_forward, bb._run_backward
def rec():
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
def f(*a, **k):
    rec(); r = of(*a, **k); rec(); return r
def b(*a, **k):
    rec(); r = ob(*a, **k); rec(); return r
object.__setattr__(bb, "_run_forward", f); object.__setattr__(bb, "_run_backward", b)
def step():
    out = model(inputs)
    loss, _ = blender.computeLoss(out, events, durations)
    loss.backward()
    opt.step(); sched.step(); opt.zero_grad()
for _ in range(10):
    step()
torch.cuda.synchronize(); marks.clear()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
seg = {"fwd": 0.0, "tail": 0.0, "bwd": 0.0, "post": 0.0}
for i in range(K):
    m = marks[4 * i: 4 * i + 4]
    seg["fwd"] += m[0].elapsed_time(m[1]); seg["tail"] += m[1].elapsed_time(m[2]); seg["bwd"] += m[2].elapsed_time(m[3])
    if i + 1 < K:
        seg["post"] += m[3].elapsed_time(marks[4 * i + 4])
print({k: round(v / K, 3) for k, v in seg.items()}, "host enqueue ms/step", round(th / K * 1e3, 3), "wall ms/step", round(tw / K * 1e3, 3))
