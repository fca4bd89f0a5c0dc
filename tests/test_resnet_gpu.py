"""r3d_18 (reference models/resnet.py:5-227) on the HIP kernels: golden vectors from the reference's own class,
every gradient against the fp64 oracle, eval mode, dropout semantics, checkpoint compatibility."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import synth
from tests._util import load_golden, rel_err, synth_sd

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(dropout=0.0):
    from mmnn_sts_amd.models.resnet import r3d_18
    m = r3d_18(2)
    m.load_state_dict(synth_sd(R.resnet18_schema(2), "r3d."), strict=True)
    m.dropout.p = dropout
    return m.to(DEV)


@pytest.mark.parametrize("tag,shape", [("a", (2, 1, 16, 64, 64)), ("b", (3, 1, 9, 40, 52))])
def test_r3d18_train_step_golden_and_fp64(tag, shape):
    g = load_golden("g10_r3d18.npz")
    m = _model().train()
    x = torch.from_numpy(synth.uniform(f"r3d/x/{tag}", shape))
    y = m(x.to(DEV))
    cot = torch.from_numpy(synth.uniform(f"r3d/cot/{tag}", tuple(y.shape)))
    (y * cot.to(DEV)).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), g[f"{tag}/out"]) < 1e-4                  # north-star bar on the outputs
    sd_dev = m.state_dict()
    for k, v in zip(g[f"{tag}/running_names"], g[f"{tag}/running_chk"]):
        t = sd_dev[str(k)].double()
        np.testing.assert_allclose([t.sum().item(), t.abs().sum().item()], v, rtol=1e-4, atol=1e-6)
    assert int(sd_dev["stem.1.num_batches_tracked"]) == 1 and int(sd_dev["layer4.1.conv2.1.num_batches_tracked"]) == 1
    # every gradient against the oracle evaluated in fp64
    sd = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in synth_sd(R.resnet18_schema(2), "r3d.").items()}
    y64 = R.resnet18_forward(sd, x.double(), True)
    (y64 * cot.double()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), y64.detach().numpy()) < 1e-4
    gl2 = float(torch.sqrt(sum((v.grad ** 2).sum() for v in sd.values() if v.is_floating_point() and v.grad is not None)))
    bad = []
    for k, p in m.named_parameters():
        ref = sd[k].grad
        err = float((p.grad.double().cpu() - ref).norm())
        if err > 2e-3 * float(ref.norm()) + 2e-5 * gl2:
            bad.append((k, err, float(ref.norm())))
    assert not bad, (len(bad), gl2, bad[:6])
    assert len(list(m.named_parameters())) == 65
    # gradients of a few tensors also against the reference's own fp32 numbers (loosely: ReLU branch flips)
    for k in ("fc.weight", "fc.bias"):
        np.testing.assert_allclose(dict(m.named_parameters())[k].grad.cpu().numpy(), g[f"{tag}/grad/{k}"], rtol=2e-3, atol=1e-6)
    m.eval()
    with torch.no_grad():
        assert rel_err(m(x.to(DEV)).cpu().numpy(), g[f"{tag}/eval_out"]) < 1e-4


def test_r3d18_dropout_and_errors():
    m = _model(dropout=0.5).train()
    x = torch.randn(2, 1, 8, 32, 32, device=DEV)
    y1, y2 = m(x), m(x)
    assert torch.isfinite(y1).all() and not torch.equal(y1, y2)                          # fresh element-wise masks per call
    y1.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))                                                   # dropout is off in eval mode
    with pytest.raises(ValueError):
        m(torch.randn(2, 2, 8, 32, 32, device=DEV))
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 1, 8, 32, 32))


def test_direct_conv_matches_torch_reference():
    """The generic direct convolution (forward, data gradient, weight gradient) against torch's CPU conv3d in fp64 for the
    kernel / stride / padding combinations r3d_18 uses and a few irregular ones."""
    from mmnn_sts_amd import ops
    cases = [((2, 1, 5, 20, 22), 64, (1, 7, 7), (1, 2, 2), (1, 3, 3)), ((2, 64, 6, 9, 10), 8, (3, 3, 3), 1, 1),
             ((1, 8, 7, 11, 9), 16, (3, 3, 3), 2, 1), ((2, 8, 7, 11, 9), 16, (1, 1, 1), 2, 0), ((3, 5, 4, 6, 7), 19, (2, 3, 1), (1, 2, 3), (1, 0, 2))]
    for xs, co, k, s, p in cases:
        x = torch.from_numpy(synth.uniform(f"conv/x/{xs}", xs)).double().requires_grad_(True)
        w = torch.from_numpy(synth.uniform(f"conv/w/{xs}", (co, xs[1]) + tuple(k), 0.3)).double().requires_grad_(True)
        y = torch.nn.functional.conv3d(x, w, None, stride=s, padding=p)
        cot = torch.from_numpy(synth.uniform(f"conv/c/{xs}", tuple(y.shape))).double()
        (y * cot).sum().backward()
        xg = x.detach().float().to(DEV).requires_grad_(True)
        wg = w.detach().float().to(DEV).requires_grad_(True)
        yg = ops.Conv3dDirect.apply(xg, wg, s, p)
        (yg * cot.float().to(DEV)).sum().backward()
        assert tuple(yg.shape) == tuple(y.shape)
        assert rel_err(yg.detach().cpu().numpy(), y.detach().numpy()) < 2e-5, (xs, k)
        assert rel_err(xg.grad.cpu().numpy(), x.grad.numpy()) < 2e-5, (xs, k)
        assert rel_err(wg.grad.cpu().numpy(), w.grad.numpy()) < 2e-5, (xs, k)
