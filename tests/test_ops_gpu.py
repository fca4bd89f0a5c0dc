"""Edge cases of the C-ABI ops through the Python mirror: error contract, ragged / tiny shapes, dropout semantics of the MLP,
optimizer parity, per-op gradient checks against torch autograd on the CPU."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import synth
from tests._util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_error_contract():
    from mmnn_sts_amd.models.densenet import DenseNet
    from mmnn_sts_amd.models.mlp import MLP
    m = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=4, block_config=(2, 2)).to(DEV)
    with pytest.raises(ValueError, match="expected"):
        m.backbone(torch.zeros(1, 2, 32, 32, 32, device=DEV))           # wrong channel count
    with pytest.raises(ValueError, match="too small"):
        m.backbone(torch.zeros(1, 1, 4, 4, 4, device=DEV))              # no voxels left in block 2
    mlp = MLP(8, 2, 4).to(DEV).train()
    with pytest.raises(ValueError, match="more than 1 value"):
        mlp(torch.zeros(1, 8, device=DEV))                              # BatchNorm1d needs N > 1 in training (torch raises too)
    mlp.eval()
    assert mlp(torch.zeros(1, 8, device=DEV)).shape == (1, 2)


@pytest.mark.parametrize("n,c,dhw,f", [(1, 8, (1, 1, 1), 3), (3, 40, (2, 3, 5), 12), (2, 1024, (4, 4, 4), 12)])
def test_gap_linear_vs_torch(n, c, dhw, f):
    from mmnn_sts_amd import ops
    h = torch.from_numpy(synth.uniform("gap/h", (n, c) + dhw)).requires_grad_(True)
    w = torch.from_numpy(synth.uniform("gap/w", (f, c), 0.2)).requires_grad_(True)
    b = torch.from_numpy(synth.uniform("gap/b", (f,), 0.1)).requires_grad_(True)
    cot = torch.from_numpy(synth.uniform("gap/cot", (n, f)))
    ref = torch.nn.functional.linear(torch.relu(h.double()).mean(dim=(2, 3, 4)), w.double(), b.double())
    (ref * cot.double()).sum().backward()
    hg, wg, bg = (t.detach().to(DEV).requires_grad_(True) for t in (h, w, b))
    out = ops.GapLinear.apply(hg, wg, bg, 0.0, True)
    (out * cot.to(DEV)).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    for got, want in ((hg.grad, h.grad), (wg.grad, w.grad), (bg.grad, b.grad)):
        assert rel_err(got.cpu().numpy(), want.numpy()) < 1e-5


def test_mlp_dropout_drops_whole_rows():
    """nn.Dropout1d on a 2-D input zeroes whole patients: with p = 0.5 a dropped row is all zeros after layer 0."""
    from mmnn_sts_amd.models.mlp import MLP
    torch.manual_seed(1)
    m = MLP(16, 2, 12, dropout_prob=0.5).to(DEV).train()
    x = torch.randn(256, 16, device=DEV)
    y = m.backbone(x)
    dead = (y.abs().sum(dim=1) == 0).float().mean().item()
    assert 0.35 < dead < 0.65         # the last layer's row dropout (drop -> relu) zeroes ~p of the patients entirely
    m2 = MLP(16, 2, 12, dropout_prob=0.5).to(DEV).eval()
    assert (m2.backbone(x).abs().sum(dim=1) == 0).float().mean().item() < 0.1


def test_fused_sgd_matches_torch_sgd():
    from mmnn_sts_amd.models.densenet import DenseNet
    from mmnn_sts_amd.optim import FusedSGD
    torch.manual_seed(0)
    a = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=4, block_config=(2, 2)).to(DEV).train()
    b = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=4, block_config=(2, 2)).to(DEV).train()
    b.load_state_dict(a.state_dict())
    oa = FusedSGD(a, lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
    x = torch.randn(2, 1, 24, 24, 24, device=DEV)
    for _ in range(3):
        for net, opt in ((a, oa), (b, ob)):
            net(x).square().sum().backward()
            opt.step()
            opt.zero_grad()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        d = float((pa.detach() - pb.detach()).abs().max())
        assert d < 2e-5 * max(1e-2, float(pb.detach().abs().max())), (k, d)    # BN betas / biases feeding a BN stay ~0


def test_state_dict_round_trip_and_device_moves():
    from mmnn_sts_amd.models.densenet import TinyDensenet
    m = TinyDensenet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12).to(DEV).eval()
    x = torch.randn(1, 2, 32, 32, 32, device=DEV)
    with torch.no_grad():
        y0 = m(x)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    m2 = TinyDensenet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12)
    m2.load_state_dict(sd)
    m2 = m2.to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(m2(x), y0)
    m2 = m2.cpu().to(DEV)                 # .to() re-allocates every tensor: the flat storage is rebuilt lazily
    with torch.no_grad():
        assert torch.equal(m2(x), y0)
    assert m2.backbone._storage_ok(full=True)


def test_versioned_weight_repack():
    """`repack_policy = "versioned"`: the weight panels are re-packed only after a parameter changed (in-place op on a Parameter,
    load_state_dict, FusedSGD), never spuriously skipped; "always" (default) repacks on every forward."""
    from mmnn_sts_amd import _lib
    from mmnn_sts_amd.models.densenet import DenseNet
    from mmnn_sts_amd.optim import FusedSGD
    torch.manual_seed(5)
    m = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, block_config=(2, 2), dropout_prob=0.0).to("cuda")
    bb = m.backbone
    x = torch.randn(2, 1, 32, 32, 32, device="cuda")
    packs = lambda: _lib.lib().mmnn_densenet_ws_offset(next(iter(bb._plans.values()))["plan"], b"#pack_launches", 0, 0)
    m.eval()
    with torch.no_grad():
        y0 = bb(x)
        assert packs() == 1
        bb(x)
        assert packs() == 2                                   # default policy: every forward
        bb.repack_policy = "versioned"
        y1 = bb(x); n = packs()
        y2 = bb(x)
        assert packs() == n and torch.equal(y1, y0) and torch.equal(y2, y0)      # unchanged parameters: no repack, same result
        bb.conv0.weight.mul_(1.5)                             # in-place op on a Parameter: seen through its version counter
        y3 = bb(x)
        assert packs() == n + 1 and not torch.equal(y3, y0)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["backbone.denseblock1.denselayer1.layers.conv2.weight"] *= 0.5
        m.load_state_dict(sd)
        y4 = bb(x)
        assert packs() == n + 2 and not torch.equal(y4, y3)
    m.train()
    opt = FusedSGD(m, lr=0.1)
    bb(x).sum().backward()
    before = packs()
    opt.step()                                                # raw-pointer update -> mark_params_changed()
    with torch.no_grad():
        y5 = bb(x)
    assert packs() == before + 1
    m.eval()
    with torch.no_grad():
        ref = bb(x); bb.repack_policy = "always"; again = bb(x)
    assert torch.equal(ref, again)


def _tiny_fusion(blend=True, seed=0):
    from mmnn_sts_amd.models.densenet import DenseNet
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    torch.manual_seed(seed)
    img = DenseNet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, block_config=(2, 2), dropout_prob=0.0)
    mm = MultiModalModel(img, [f"p{i}" for i in range(8)], 2, 12, blend=blend)
    for m in mm.modules():
        if m.__class__.__name__.startswith("Dropout"):
            m.p = 0.0
    return mm


def test_fused_sgd_under_onecycle_matches_torch_sgd():
    """main.py:410-414: SGD(momentum, nesterov, weight_decay) driven by OneCycleLR, which cycles BOTH `lr` and `momentum`
    (0.95 -> 0.85 -> 0.95) through param_groups on every step.  FusedSGD (flat backbone launch + multi-tensor launch for the MLP /
    feature layer / heads) against torch.optim.SGD under the same scheduler, 6 steps of the fusion model, every parameter incl. the tail
    tensors; the never-trained tensors (no gradient: SURVEY A6) must stay bit-identical to their initial values in both."""
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    from mmnn_sts_amd.losses.losses import CoxPH
    from mmnn_sts_amd.optim import FusedSGD
    from mmnn_sts_amd.utils.utils import surv_criterion
    a, b = _tiny_fusion().to(DEV).train(), _tiny_fusion().to(DEV).train()
    b.load_state_dict(a.state_dict())
    init = {k: v.detach().clone() for k, v in a.named_parameters()}
    oa = FusedSGD(a, lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-3)
    ob = torch.optim.SGD(b.parameters(), lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-3)
    steps = 6
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=0.05, total_steps=steps)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=0.05, total_steps=steps)
    g = torch.Generator().manual_seed(3)
    x = {"image": torch.randn(4, 2, 24, 24, 24, generator=g).to(DEV), "clinical": torch.randn(4, 8, generator=g).to(DEV)}
    ev = torch.tensor([[1, 0], [0, 1], [1, 1], [1, 0]], device=DEV)
    du = torch.tensor([[100, 250], [300, 50], [20, 400], [75, 60]], device=DEV)
    moms = []
    for _ in range(steps):
        for net, opt, sch in ((a, oa, sa), (b, ob, sb)):
            gb = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
            loss, _ = gb.computeLoss(net(x), ev, du)
            loss.backward()
            opt.step()
            sch.step()
            opt.zero_grad()
        moms.append((oa.param_groups[0]["momentum"], ob.param_groups[0]["momentum"], oa.param_groups[0]["lr"], ob.param_groups[0]["lr"]))
    assert all(m[0] == m[1] and m[2] == m[3] for m in moms) and len({round(m[0], 6) for m in moms}) > 2     # momentum really cycled
    untouched = 0
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        d = float((pa.detach() - pb.detach()).abs().max())
        assert d < 5e-5 * max(1e-2, float(pb.detach().abs().max())), (k, d)
        if torch.equal(pb.detach(), init[k]):
            assert torch.equal(pa.detach(), init[k]), k           # no gradient => no weight decay either, in both optimizers
            untouched += 1
    assert untouched == 4                                          # class_layers.out.{weight,bias}, clinical output_head.dense6.{weight,bias}
    moved = sum(1 for k, p in a.named_parameters() if not torch.equal(p.detach(), init[k]))
    assert moved == len(init) - 4


def test_eval_forward_of_frozen_encoder_lets_heads_train():
    """an eval-mode forward keeps no activations, so a backward INTO the backbone must fail loudly -- but a fully frozen
    encoder (the legitimate 'train heads on a frozen eval-mode encoder' use, which plain PyTorch and the reference allow) has no gradient
    to produce: its output is a plain tensor and the layers on top train normally."""
    from mmnn_sts_amd.models.densenet import DenseNet
    torch.manual_seed(2)
    m = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=4, block_config=(2, 2)).to(DEV)
    x = torch.randn(2, 1, 24, 24, 24, device=DEV)
    m.eval()
    with pytest.raises(RuntimeError, match="eval-mode"):
        m(x).sum().backward()                                      # backbone parameters require grad: loud failure
    for p in m.backbone.parameters():
        p.requires_grad_(False)
    m.features.train(); m.class_layers.train()
    y = m(x)
    y.square().sum().backward()
    assert m.class_layers.out.weight.grad is not None and m.features.feature_layer.weight.grad is not None
    assert all(p.grad is None for p in m.backbone.parameters())
    with pytest.raises(RuntimeError, match="eval-mode"):
        m(x.requires_grad_(True)).sum().backward()                 # a gradient wrt the INPUT would need the activations too


def test_cox_blend_typed_operands_and_backward_scale():
    """The Cox kernel reads int64 / int32 / float32 / float64 / bool sort keys and weights directly (no conversion pass), and the
    autograd adjoint scales the saved gradient in a HIP kernel: d(3 * loss + sum_h a_h * head_loss_h) must match the fp64 oracle."""
    from mmnn_sts_amd import ops
    preds = torch.from_numpy(synth.uniform("coxt/p", (3, 6, 2)))
    ev = torch.tensor([[1, 0], [0, 1], [1, 1], [1, 0], [0, 0], [1, 1]])
    du = torch.tensor([[100, 250], [300, 50], [20, 400], [75, 75], [5, 900], [60, 61]])
    hw = torch.tensor([0.5, 0.3, 0.2])
    coef = torch.tensor([0.7, -1.1, 0.4])
    ref_p = preds.double().requires_grad_(True)
    heads = torch.stack([sum(R.pycox_cox_ph_loss(ref_p[h][:, c], ev[:, c].double(), du[:, c].double()) for c in range(2)) for h in range(3)])
    (3.0 * (hw.double() * heads).sum() + (coef.double() * heads).sum()).backward()
    outs = []
    for kd, wd in ((torch.int64, torch.int64), (torch.float32, torch.float32), (torch.bool, torch.int32), (torch.float64, torch.float64),
                   (torch.int16, torch.float16)):
        p = preds.to(DEV).requires_grad_(True)
        loss, hl = ops.CoxBlend.apply(p, ev.to(kd).to(DEV), du.to(wd).to(DEV), hw.to(DEV))
        (3.0 * loss + (coef.to(DEV) * hl).sum()).backward()
        assert rel_err(hl.detach().cpu().numpy(), heads.detach().numpy()) < 1e-5
        assert rel_err(p.grad.cpu().numpy(), ref_p.grad.numpy()) < 2e-5
        outs.append(p.grad.cpu())
    assert all(torch.equal(outs[0], o) for o in outs[1:])          # the element type of the operands changes nothing
    p = preds.to(DEV).requires_grad_(True)
    loss, hl = ops.CoxBlend.apply(p, ev.to(DEV), du.to(DEV), hw.to(DEV))
    hl[1].backward()                                               # only a head loss in the graph (dloss is None)
    ref2 = preds.double().requires_grad_(True)
    sum(R.pycox_cox_ph_loss(ref2[1][:, c], ev[:, c].double(), du[:, c].double()) for c in range(2)).backward()
    assert rel_err(p.grad.cpu().numpy(), ref2.grad.numpy()) < 2e-5
