"""Developer tool: host-side enqueue time per training step vs device time (is the step launch-bound?)."""
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
def step():
    out = model(x); loss, _ = bl.computeLoss(out, ev, du); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step, wall {1e3*(t2-t0)/20:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
