"""Debug helper (not a test): per-parameter gradient error table for one backbone case."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import restatement as R, synth
from tests._util import synth_sd
from tests._native import NativeBackbone

in_ch, blocks, dhw, n = 2, (6, 12, 24, 16), (64, 64, 64), 2
if len(sys.argv) > 1:
    in_ch, n = int(sys.argv[1]), int(sys.argv[2]); dhw = tuple(int(v) for v in sys.argv[3:6]); blocks = tuple(int(v) for v in sys.argv[6:])
cfg = R.DenseNetCfg(in_channels=in_ch, block_config=blocks)
sch = R.densenet_schema(cfg)
sd = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in synth_sd(sch, "densenet.").items()}
x = torch.from_numpy(synth.uniform(f"bb/{n}x{in_ch}x{dhw}", (n, in_ch) + dhw))
h = R.densenet_backbone(sd, x.double(), cfg, True)
cot = torch.from_numpy(synth.uniform("bb/cot", tuple(h.shape)))
(h * cot.double()).sum().backward()
nb = NativeBackbone(cfg, n, *dhw)
flat, run = nb.flatten(synth_sd(sch, "densenet."))
out = nb.forward(flat, run, x.cuda(), True)
g = nb.backward(flat, x.cuda(), cot.cuda())
torch.cuda.synchronize()
got = nb.unflatten(g.cpu())
for k in reversed(list(got)):
    ref = sd[k].grad
    e = float((got[k].double() - ref).norm()); r = float(ref.norm())
    print(f"{e / max(r, 1e-30):9.2e} {r:10.3e} {k}")
