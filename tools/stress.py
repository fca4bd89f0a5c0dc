"""Developer tool: repeat forward+backward of one backbone case and compare every run with the first (race detector)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from tests._native import NativeBackbone
from tests._util import synth_sd

in_ch, s, n, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = R.DenseNetCfg(in_channels=in_ch)
nb = NativeBackbone(cfg, n, s, s, s)
flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
x = torch.randn(n, in_ch, s, s, s, device="cuda")
cot = torch.randn(nb.out_shape, device="cuda")
# background load on another stream to create uneven timing
bg = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
ref_out = ref_g = None
bad = 0
names = nb.unflatten(torch.zeros(nb.n_params))
for it in range(iters):
    if it % 3 == 1:
        with torch.cuda.stream(bg):
            for _ in range(3):
                junk = junk * 1.0001 + 0.1
    out = nb.forward(flat, run.clone(), x, True, seed=1)
    g = nb.backward(flat, x, cot, seed=1)
    torch.cuda.synchronize()
    if ref_out is None:
        ref_out, ref_g = out.clone(), g.clone()
        continue
    eo = float((out - ref_out).abs().max() / ref_out.abs().max())
    eg = float((g - ref_g).abs().max() / ref_g.abs().max())
    if eo > 1e-5 or eg > 1e-4:
        bad += 1
        off = 0
        worst = []
        for k, v in names.items():
            m = v.numel()
            d = float((g[off:off + m] - ref_g[off:off + m]).abs().max())
            r = float(ref_g[off:off + m].abs().max())
            if d > 1e-4 * max(r, 1e-3):
                worst.append((k, d, r))
            off += m
        print(f"iter {it}: out dev {eo:.2e}, grad dev {eg:.2e}; {len(worst)} tensors differ; first: {worst[:2]} last: {worst[-2:]}", flush=True)
print(f"done: {bad} deviating runs of {iters - 1}")
