"""Per-phase shader-clock stamps of every convolution launch of one forward + backward (developer tool; FpropArgs::trace).
   python tools/phase_trace.py [N S]      prints, per launch: kernel tag, grid, and the median over traced blocks of the phase spans."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from tests._native import NativeBackbone
from tests._util import synth_sd
from mmnn_sts_amd import _lib

n, s = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 128)
cfg = R.DenseNetCfg(in_channels=2)
nb = NativeBackbone(cfg, n, s, s, s, dropout=0.2)
flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
x = torch.randn(n, 2, s, s, s, device="cuda")
cot = torch.randn(nb.out_shape, device="cuda")
L = _lib.lib()
_lib.check(L.mmnn_densenet_set_option(nb.plan, b"single_stream", 1), "opt")
os.environ.setdefault("MMNN_NO_WGRAD_BATCH", "1")        # per-layer weight-gradient launches carry the trace pointer
for _ in range(3):
    nb.forward(flat, run, x, True, seed=1); nb.backward(flat, x, cot, seed=1)
torch.cuda.synchronize()
SLOTS = 400
buf = torch.zeros(SLOTS * 64 * 16, dtype=torch.int64, device="cuda")
_lib.check(L.mmnn_densenet_set_option(nb.plan, b"trace_slots", SLOTS), "opt")
_lib.check(L.mmnn_densenet_set_option(nb.plan, b"trace_buffer", buf.data_ptr()), "opt")
nb.forward(flat, run, x, True, seed=1); nb.backward(flat, x, cot, seed=1)
torch.cuda.synchronize()
_lib.check(L.mmnn_densenet_set_option(nb.plan, b"trace_buffer", 0), "opt")
t = buf.cpu().numpy().astype(np.uint64).reshape(SLOTS, 64, 16)
names = ["load0 issue", "prologue", "1st store+sync", "chunk loop", "KS reduce", "kz publish+ticket", "kz sum", "epilogue", "stats+end"]
order = [0, 1, 2, 9, 3, 4, 5, 6, 7, 8]     # stamp ids in time order
print("seq  taps pro epi    M  Cin  grid(x,y,z) waves KC | total | " + " | ".join(names))
for i in range(SLOTS):
    tag = int(t[i, 0, 10])
    if tag == 0:
        continue
    taps, pro, epi, M, cin = tag >> 48, (tag >> 40) & 255, (tag >> 32) & 255, (tag >> 16) & 0xFFFF, tag & 0xFFFF
    g = int(t[i, 0, 11]); gx, gy, gz = g >> 32, (g >> 16) & 0xFFFF, g & 0xFFFF
    w = int(t[i, 0, 12]); waves, kc = w >> 32, (w >> 16) & 0xFFFF
    if pro == 9:      # wgrad3: per-block cycle SUMS over the block's tiles
        blk = np.array([[int(t[i, b, k]) for k in range(7)] for b in range(16) if int(t[i, b, 6]) > 0], dtype=np.float64)
        if len(blk):
            m = np.median(blk, axis=0)
            print(f"{i:3d}  wgrad{3 if taps == 27 else 1} M {M} Cin {cin} grid {gx},{gy},{gz}: tiles/block {m[6]:.0f} | first load+coef {m[0]:.0f} | per tile: store {m[1] / m[6]:.0f}  "
                  f"barrier {m[2] / m[6]:.0f}  load issue {m[3] / m[6]:.0f}  mfma {m[4] / m[6]:.0f}  barrier {m[5] / m[6]:.0f}  = {(m[1:6].sum()) / m[6]:.0f} cycles")
        continue
    rows, head = [], []
    for b in range(64):
        st = t[i, b]
        if st[0] == 0 or st[8] == 0:
            continue          # block did not run to the end (not the last K slice) or slot unused
        seq = [int(st[k]) for k in order]
        if any(v == 0 for v in seq):
            # phases that were skipped (e.g. no kz): carry the previous stamp forward
            for j in range(1, len(seq)):
                if seq[j] == 0:
                    seq[j] = seq[j - 1]
        rows.append(np.diff(np.array(seq, dtype=np.float64)))
        # inside "load0 issue": argument warm-up done / statistic loads issued / first weight loads issued, as offsets from the start
        head.append([float(int(st[k]) - int(st[0])) if st[k] else 0.0 for k in (13, 14, 15)])
    if not rows:
        continue
    d = np.median(np.stack(rows), axis=0)
    h = np.median(np.stack(head), axis=0)
    print(f"{i:3d}  {taps:4d} {pro:3d} {epi:3d} {M:4d} {cin:4d}  {gx:4d},{gy:2d},{gz:1d} {waves:5d} {kc:3d} | {d.sum():7.0f} | " + " | ".join(f"{v:7.0f}" for v in d)
          + " || args %5.0f  stats issued %5.0f  weights issued %5.0f" % tuple(h))
