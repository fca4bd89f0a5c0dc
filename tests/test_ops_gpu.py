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
