"""Repeat F / B on one plan; on the first forward whose output deviates from call 1, list which workspace regions differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R
from tests._native import NativeBackbone
from tests._util import synth_sd
in_ch, s, n = 1, 64, 2
seq = os.environ.get("SEQ", "FFBFBFFBBF")
cfg = R.DenseNetCfg(in_channels=in_ch)
nb = NativeBackbone(cfg, n, s, s, s)
flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
x = torch.randn(n, in_ch, s, s, s, device="cuda")
cot = torch.randn(nb.out_shape, device="cuda")
regs = [("pk_conv0", 0, 0, (7 * 56 * 64,), torch.float32), ("conv0", 0, 0, (n, 64, 32, 32, 32), torch.float32),
        ("st_conv0", 0, 0, (2, 8, 64), torch.float64)]
dims, c = 16, 64
for b, nl in enumerate(cfg.block_config):
    ct = c + 32 * nl
    regs.append(("st_x", b, 0, (2, 8, ct), torch.float64))
    regs.append(("x", b, 0, (n, ct, dims ** 3), torch.float32))
    for l in range(nl):
        regs.append(("pk_c1", b, l, (c + 32 * l, 128), torch.float32))
        regs.append(("t1", b, l, (n, 128, dims ** 3), torch.float32))
        regs.append(("st_t1", b, l, (2, 8, 128), torch.float64))
        regs.append(("pk_c2f", b, l, (27 * 128 * 32,), torch.float32))
    if b < 3:
        regs.append(("ap", b, 0, (n, ct, (dims // 2) ** 3), torch.float32))
    c = ct // 2; dims //= 2
ref_ws = torch.empty_like(nb.ws)
ref_o = ref_g = None
def report():
    shown = 0
    for name, i, j, shape, dt in regs:
        a = nb.region(name, shape, i, j, dt)
        off = nb.L.mmnn_densenet_ws_offset(nb.plan, name.encode(), i, j)
        r = ref_ws[off:off + a.numel() * a.element_size()].view(dt).view(shape)
        d = (a.double() - r.double()).abs().nan_to_num(nan=1e30)
        if float(d.max()) > 0:
            extra = ""
            if name == "x":
                bad = (d.amax(dim=(0, 2)) > 0).nonzero().flatten()
                extra = f" bad channels {bad[:4].tolist()}..{bad[-1].item()} ({bad.numel()})"
            print(f"    {name}[{i}][{j}] dev {float(d.max()):.3g} nan {int(torch.isnan(a).sum())} ref-nan {int(torch.isnan(r).sum())}{extra}")
            shown += 1
            if shown >= 12: break
for k, op in enumerate(seq):
    if op == "F":
        o = nb.forward(flat, run.clone(), x, True, seed=1); torch.cuda.synchronize()
        if ref_o is None:
            ref_o = o; ref_ws.copy_(nb.ws); torch.cuda.synchronize(); print(k, "F ref")
        else:
            d = float((o - ref_o).abs().nan_to_num(nan=1e30).max()); print(k, "F dev", d)
            if d > 0: report()
    else:
        g = nb.backward(flat, x, cot, seed=1); torch.cuda.synchronize()
        if ref_g is None:
            ref_g = g; print(k, "B ref", "nan", int(torch.isnan(g).sum()))
        else:
            print(k, "B dev", float((g - ref_g).abs().nan_to_num(nan=1e30).max()), "nan", int(torch.isnan(g).sum()))
