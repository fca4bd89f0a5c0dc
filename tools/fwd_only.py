"""Forward-only timing before / after the multi-stream backward has run (developer experiment)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from tests._native import NativeBackbone
from tests._util import synth_sd
n, s = 2, 128
cfg = R.DenseNetCfg(in_channels=2)
nb = NativeBackbone(cfg, n, s, s, s, dropout=0.2)
flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
x = torch.randn(n, 2, s, s, s, device="cuda")
cot = torch.randn(nb.out_shape, device="cuda")
grad = torch.zeros_like(flat)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def fwd_time(k=10):
    t = 0.0
    for _ in range(k):
        ev[0].record(); nb.forward(flat, run, x, True, seed=1); ev[1].record(); torch.cuda.synchronize()
        t += ev[0].elapsed_time(ev[1])
    return t / k
for _ in range(3): nb.forward(flat, run, x, True, seed=1)
torch.cuda.synchronize()
print("forward-only, no backward yet: %.3f ms" % fwd_time())
nb.backward(flat, x, cot, grad=grad, seed=1); torch.cuda.synchronize()
print("forward-only, after one backward: %.3f ms" % fwd_time())
for _ in range(10):
    nb.forward(flat, run, x, True, seed=1); nb.backward(flat, x, cot, grad=grad, seed=1)
torch.cuda.synchronize()
print("forward-only, after 10 F+B: %.3f ms" % fwd_time())
t = 0.0
for _ in range(10):
    ev[0].record(); nb.forward(flat, run, x, True, seed=1); ev[1].record()
    nb.backward(flat, x, cot, grad=grad, seed=1); torch.cuda.synchronize()
    t += ev[0].elapsed_time(ev[1])
print("forward inside F+B loop: %.3f ms" % (t / 10))
