"""Time the native backbone forward+backward (developer tool; not the bench contract)."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from tests._native import NativeBackbone
from tests._util import synth_sd

n, s, iters = 2, 128, 5
if len(sys.argv) > 1:
    n, s, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = R.DenseNetCfg(in_channels=2)
nb = NativeBackbone(cfg, n, s, s, s, dropout=0.2)
flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
x = torch.randn(n, 2, s, s, s, device="cuda")
cot = torch.randn(nb.out_shape, device="cuda")
grad = torch.zeros_like(flat)
torch.cuda.synchronize()
if os.environ.get('OWN_STREAM') == '1':
    torch.cuda.set_stream(torch.cuda.Stream())
for _ in range(2):
    nb.forward(flat, run, x, True, seed=1)
    nb.backward(flat, x, cot, grad=grad, seed=1)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
pause = float(os.environ.get('PAUSE_MS', '0')) / 1e3
for _ in range(iters):
    if pause: time.sleep(pause)
    ev[0].record()
    nb.forward(flat, run, x, True, seed=1)
    ev[1].record()
    nb.backward(flat, x, cot, grad=grad, seed=1)
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1])
    tb += ev[1].elapsed_time(ev[2])
print(f"N={n} S={s}: forward {tf / iters:.3f} ms, backward {tb / iters:.3f} ms, total {(tf + tb) / iters:.3f} ms -> {n / ((tf + tb) / iters) * 1e3:.1f} volumes/s")
