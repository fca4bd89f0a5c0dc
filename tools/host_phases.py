"""Host-side wall time of the pieces of a training step (developer tool): where does the enqueueing thread block?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmnn_sts_amd.losses.GradientBlender import GradientBlender
from mmnn_sts_amd.losses.losses import CoxPH
from mmnn_sts_amd.optim import FusedSGD
from mmnn_sts_amd.utils.utils import surv_criterion
dev = torch.device("cuda:0")
model = bench.build_model(dev).train()
opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
bl = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
x, ev, du = bench.synth_batch(dev, 0, 2, int(os.environ.get("S", 128)))
bb = model.image_model.model.backbone
T = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
    return w
object.__setattr__(bb, "_run_forward", timed("backbone_forward_call", bb._run_forward))
object.__setattr__(bb, "_run_backward", timed("backbone_backward_call", bb._run_backward))
def step():
    t0 = time.perf_counter(); out = model(x); t1 = time.perf_counter()
    loss, _ = bl.computeLoss(out, ev, du); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    opt.step(); opt.zero_grad(); t4 = time.perf_counter()
    for k, v in (("model()", t1 - t0), ("computeLoss", t2 - t1), ("loss.backward()", t3 - t2), ("opt", t4 - t3)):
        T[k] = T.get(k, 0.0) + v
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print({k: round(v / K * 1e3, 3) for k, v in T.items()}, "host ms/step", round(th / K * 1e3, 3), "wall", round(tw / K * 1e3, 3))
