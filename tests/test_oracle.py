"""The CPU restatement (oracle/restatement.py) against the golden vectors produced from the reference's own classes.

These are `not gpu` tests: they pin the ORACLE.  The HIP path is compared with the oracle in the `-m gpu` tests.
Tolerances: the reference's own fp32 result moves by <= 5.4e-7 between thread counts, so 2e-5 relative
on outputs / losses is a tight but safe bound for an identical-arithmetic restatement.
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import synth
from tests._util import N_CLIN, clin_in, image_in, labels, load_golden, rel_err, stat3, synth_sd

RTOL = 2e-5


def test_schema_counts():
    cfg = R.DenseNetCfg()
    s = R.multimodal_schema(cfg, N_CLIN, 2, 12)
    assert len(s) == 779
    n_param = sum(int(np.prod(v)) for k, v in s.items() if "running" not in k and "num_batches" not in k)
    assert n_param == 11_279_170
    assert sum(int(np.prod(v)) for k, v in R.densenet_schema(cfg).items()
               if "running" not in k and "num_batches" not in k) == 11_276_902
    tiny = R.DenseNetCfg(block_config=(6, 12, 4))
    assert sum(int(np.prod(v)) for k, v in R.densenet_schema(tiny).items()
               if "running" not in k and "num_batches" not in k) == 3_398_118


def test_cox_known_answers():
    g = load_golden("g6_blender.npz")
    k1 = R.CoxPH(torch.tensor([.3, -.2, .1, .4]), torch.tensor([1, 0, 1, 1]), torch.tensor([100, 250, 300, 50]))
    assert abs(k1.item() - 1.04063249) < 2e-6 and abs(k1.item() - g["kat1"][0]) < 1e-6
    intended = R.pycox_cox_ph_loss(torch.tensor([.3, -.2, .1, .4]), torch.tensor([100, 250, 300, 50]), torch.tensor([1, 0, 1, 1]))
    assert abs(intended.item() - 0.68245322) < 2e-6
    P = torch.tensor([[.3, -.2], [.1, .4], [-.5, .2], [0, .7]])
    E = torch.tensor([[1, 0], [0, 1], [1, 1], [0, 0]])
    D = torch.tensor([[100, 250], [300, 50], [20, 400], [75, 75]])
    k2 = R.surv_criterion(R.CoxPH, P, E, D)
    assert abs(k2.item() - 1.85874867) < 2e-6 and abs(k2.item() - g["kat2"][0]) < 1e-6
    preds = torch.tensor([[[.3, -.2], [.1, .4]], [[.5, 0], [-.1, .2]], [[0, .1], [.2, -.3]]])
    ev, du = labels(2)
    b = R.Blender()
    loss, sel = b.compute_loss(preds, ev, du)
    assert abs(loss.item() - 1.26084220) < 2e-6 and abs(sel.item() - 1.46317768) < 2e-6
    np.testing.assert_allclose(b.head_losses(preds, ev, du).numpy(), g["kat3_heads"], rtol=1e-6)


def test_blender_update_sequence():
    g = load_golden("g6_blender.npz")
    b = R.Blender()
    for it in range(3):
        tp = torch.from_numpy(synth.uniform(f"gb/train/{it}", (3, 12, 2)))
        vp = torch.from_numpy(synth.uniform(f"gb/val/{it}", (3, 10, 2)))
        te = torch.from_numpy((synth.uniform("gb/te", (12, 2)) > -0.2).astype(np.int64))
        ve = torch.from_numpy((synth.uniform("gb/ve", (10, 2)) > -0.2).astype(np.int64))
        td = torch.from_numpy((1 + np.floor((synth.uniform("gb/td", (12, 2)) * .5 + .5) * 2998)).astype(np.int64))
        vd = torch.from_numpy((1 + np.floor((synth.uniform("gb/vd", (10, 2)) * .5 + .5) * 2998)).astype(np.int64))
        b.update_weights(tp, te, td, vp, ve, vd)
        np.testing.assert_allclose(b.weights.numpy(), g["upd_weights"][it], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b.ltn.numpy(), g["upd_losses"][it][0], rtol=1e-6)
        np.testing.assert_allclose(b.lvn.numpy(), g["upd_losses"][it][1], rtol=1e-6)
    assert len(b.history) == 3


@pytest.mark.parametrize("n", [2, 8])
def test_mlp(n):
    g = load_golden("g1_mlp.npz")
    sch = R.mlp_schema(N_CLIN, 2, 12)
    x = clin_in(n)
    sd = synth_sd(sch, "mlp.")
    with torch.no_grad():
        out = R.mlp_forward(sd, x, False, 0.0)
    assert rel_err(out.numpy(), g[f"eval_out_n{n}"]) < RTOL
    sd = synth_sd(sch, "mlp.", requires_grad=True)
    f = R.mlp_features(sd, x, True, 0.0)
    y = torch.nn.functional.linear(f, sd["output_head.dense6.weight"], sd["output_head.dense6.bias"])
    (y * torch.from_numpy(synth.uniform("mlp/cot", tuple(y.shape)))).sum().backward()
    assert rel_err(f.detach().numpy(), g[f"train_feat_n{n}"]) < RTOL
    for k in sch:
        if "running" in k:
            assert rel_err(sd[k].numpy(), g[f"run_n{n}/{k}"]) < RTOL
        elif "num_batches" not in k:
            ref = g[f"grad_n{n}/{k}"]
            np.testing.assert_allclose(sd[k].grad.numpy(), ref, rtol=1e-3, atol=2e-6 * max(1.0, np.abs(ref).max()))


def _check_densenet(file, cfg, in_ch, s):
    g = load_golden(file)
    sch = R.densenet_schema(cfg)
    sd = synth_sd(sch, "densenet.")
    x = image_in(2, in_ch, s)
    taps = {}
    with torch.no_grad():
        h = R.densenet_backbone(sd, x, cfg, True, taps=taps)
        f = R.densenet_features(sd, h, cfg, True)
        y = torch.nn.functional.linear(f, sd["class_layers.out.weight"], sd["class_layers.out.bias"])
    assert rel_err(h.numpy(), g["norm5"]) < RTOL
    assert rel_err(f.numpy(), g["features"]) < RTOL
    assert rel_err(y.numpy(), g["out"]) < RTOL
    name_map = {"conv0": "conv0", "pool0": "stem", "norm5": "norm5"}
    for nm, st in zip(g["tap_names"], g["tap_stats"]):
        nm = str(nm)
        key = name_map.get(nm, nm.replace("denseblock", "block").replace("transition", "trans"))
        np.testing.assert_allclose(stat3(taps[key]), st, rtol=1e-5, atol=1e-7)
    run = {str(k): v for k, v in zip(g["running_names"], g["running_chk"])}
    for k, v in run.items():
        t = sd[k].double()
        np.testing.assert_allclose([t.sum().item(), t.abs().sum().item()], v, rtol=1e-5, atol=1e-6)
    assert int(sd["backbone.norm0.num_batches_tracked"]) == int(g["nbt"][0]) == 1
    with torch.no_grad():
        ye = R.densenet_forward(sd, x, cfg, False)
    assert rel_err(ye.numpy(), g["eval_out"]) < RTOL


@pytest.mark.parametrize("in_ch,s", [(1, 32), (2, 32), (2, 64)])
def test_densenet121(in_ch, s):
    _check_densenet(f"g2_densenet_in{in_ch}_s{s}.npz", R.DenseNetCfg(in_channels=in_ch), in_ch, s)


def test_tiny_densenet():
    _check_densenet("g7_tiny_in2_s32.npz", R.DenseNetCfg(in_channels=2, block_config=(6, 12, 4)), 2, 32)


@pytest.mark.parametrize("s", [32, 64])
@pytest.mark.parametrize("blend", [True, False])
def test_fusion_train_step(s, blend):
    g = load_golden(f"g3_fusion_s{s}.npz")
    tag = "blend" if blend else "plain"
    cfg = R.DenseNetCfg()
    sch = R.multimodal_schema(cfg, N_CLIN, 2, 12)
    sd = synth_sd(sch, "fusion.", requires_grad=True)
    x, c = image_in(2, 2, s), clin_in(2)
    ev, du = labels(2)
    out = R.multimodal_forward(sd, x, c, cfg, True, blend, mlp_dropout=0.0)
    if blend:
        b = R.Blender()
        loss, sel = b.compute_loss(out, ev, du)
        np.testing.assert_allclose(b.head_losses(out, ev, du).detach().numpy(), g[f"{tag}/head_losses"], rtol=RTOL)
        assert abs(sel.item() - g[f"{tag}/selection_loss"][0]) < RTOL * abs(sel.item())
    else:
        loss = R.surv_criterion(R.CoxPH, out, ev, du)
    loss.backward()
    assert rel_err(out.detach().numpy(), g[f"{tag}/out"]) < RTOL
    assert abs(loss.item() - g[f"{tag}/loss"][0]) < RTOL * abs(loss.item())
    # gradients: every parameter's L2 norm, absolute tolerance tied to the global norm
    gl2 = float(g[f"{tag}/grad_global_l2"][0])
    mine = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.requires_grad and v.grad is not None)).item()
    assert abs(mine - gl2) < 1e-3 * gl2
    for k, l2, head in zip(g[f"{tag}/grad_names"], g[f"{tag}/grad_l2"], g[f"{tag}/grad_head"]):
        k = str(k)
        if np.isnan(l2):
            assert sd[k].grad is None                       # class_layers / MLP output_head never get a gradient (A6)
            continue
        gr = sd[k].grad
        assert abs(gr.double().norm().item() - l2) < 2e-3 * l2 + 2e-5 * gl2, k
        np.testing.assert_allclose(gr.flatten()[:8].numpy(), head[: min(8, gr.numel())], rtol=5e-3, atol=2e-5 * gl2)
    with torch.no_grad():
        oe = R.multimodal_forward(sd, x, c, cfg, False, blend)
    assert rel_err(oe.numpy(), g[f"{tag}/eval_out"]) < RTOL


def test_gradcam():
    g = load_golden("g5_gradcam_s64.npz")
    cfg = R.DenseNetCfg()
    sd = synth_sd(R.multimodal_schema(cfg, N_CLIN, 2, 12), "fusion.")
    s = 64
    out, maps, small = R.multimodal_gradcam(sd, image_in(1, 2, s), clin_in(1), cfg)
    assert rel_err(out.numpy(), g["preds"]) < RTOL
    for i, m in enumerate(maps):
        np.testing.assert_allclose(m[:: s // 8, :: s // 8, :: s // 8].numpy(), g[f"map{i}_coarse"], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(stat3(m), g[f"map{i}_stats"], rtol=1e-3, atol=1e-5)


def _cls_inputs():
    n = 6
    logits = torch.from_numpy(synth.uniform("cls/logits", (3, n, 2), 2.0))
    targets = torch.from_numpy((synth.uniform("cls/targets", (n, 2)) > 0).astype(np.float32))
    return logits, targets


def _cls_update_inputs(it):
    tp = torch.sigmoid(torch.from_numpy(synth.uniform(f"cls/train/{it}", (3, 12, 2), 2.0)))
    vp = (torch.sigmoid(torch.from_numpy(synth.uniform(f"cls/val/{it}", (3, 10, 2), 2.0))) > 0.5).float()
    tt = torch.from_numpy((synth.uniform("cls/tt", (12, 2)) > 0).astype(np.float32))
    vt = torch.from_numpy((synth.uniform("cls/vt", (10, 2)) > 0).astype(np.float32))
    return tp, tt, vp, vt


def test_classification_blender_and_bce():
    """Oracle restatement of the classification branch (BCE-with-logits, blender, update sign) vs the reference (G9)."""
    g = load_golden("g9_classification.npz")
    pw = torch.from_numpy(g["pos_weight"])
    logits, targets = _cls_inputs()
    bce = lambda p, t: R.bce_with_logits(p, t, pw, "none")
    np.testing.assert_allclose(R.bce_with_logits(logits[0], targets, pw, "sum").item(), g["criterion_sum"][0], rtol=1e-6)
    for red in ("sum", "mean"):
        x = logits.clone().requires_grad_(True)
        b = R.ClassBlender(bce, red)
        loss = b.compute_loss(x, targets)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"{red}/loss"][0], rtol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), g[f"{red}/grad"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b.compute_loss(logits, targets, reduce_to_heads=True).numpy(), g[f"{red}/heads"], rtol=1e-6)
        assert len(b.history) == int(g[f"{red}/history_len"][0]) == 1
    np.testing.assert_allclose(R.ClassBlender(bce).compute_loss(logits, targets, no_reduce=True).numpy(), g["no_reduce"], rtol=1e-6)
    b = R.ClassBlender(bce)
    for it in range(3):
        b.update_weights(*_cls_update_inputs(it))
        np.testing.assert_allclose(b.weights.numpy(), g["upd_weights"][it], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(np.stack([b.ltn.numpy(), b.lvn.numpy()]), g["upd_losses"][it], rtol=1e-6)
    assert len(b.history) == int(g["upd_history_len"][0]) == 1


def test_cox_fractional_durations():
    """float32 durations / events (data/ImageDatasets.py:462 builds them with torch.Tensor([...])) must not be truncated."""
    g = load_golden("g9_classification.npz")
    for n in (4, 9):
        h = torch.from_numpy(synth.uniform(f"coxf/h{n}", (n, 2)))
        ev = torch.from_numpy((synth.uniform(f"coxf/e{n}", (n, 2)) > -0.3).astype(np.float32))
        ev[0] = 1
        du = torch.from_numpy((synth.uniform(f"coxf/d{n}", (n, 2)) * 0.5 + 0.5).astype(np.float32) * 30.0 + 0.25)
        np.testing.assert_allclose(R.surv_criterion(R.CoxPH, h, ev, du).item(), g[f"coxf/n{n}"][0], rtol=1e-6)
        np.testing.assert_allclose(R.CoxPH(h[:, 0], ev[:, 0], du[:, 0]).item(), g[f"coxf/n{n}/c0"][0], rtol=1e-6)
        assert abs(R.CoxPH(h[:, 0], ev[:, 0], du[:, 0].long().float()).item() - g[f"coxf/n{n}/c0"][0]) > 1e-4   # truncation would show


@pytest.mark.parametrize("tag,shape", [("a", (2, 1, 16, 64, 64)), ("b", (3, 1, 9, 40, 52))])
def test_r3d18(tag, shape):
    """Oracle restatement of r3d_18 (models/resnet.py:202-227) vs the reference's own class (G10)."""
    g = load_golden("g10_r3d18.npz")
    sch = R.resnet18_schema(2)
    assert len(sch) == 128
    sd = synth_sd(sch, "r3d.", requires_grad=True)
    x = torch.from_numpy(synth.uniform(f"r3d/x/{tag}", shape))
    y = R.resnet18_forward(sd, x, True)
    cot = torch.from_numpy(synth.uniform(f"r3d/cot/{tag}", tuple(y.shape)))
    (y * cot).sum().backward()
    assert rel_err(y.detach().numpy(), g[f"{tag}/out"]) < RTOL
    for k in ("fc.weight", "stem.0.weight", "layer1.0.downsample.0.weight", "layer4.1.conv2.0.weight", "layer3.0.downsample.1.bias"):
        assert rel_err(sd[k].grad.numpy(), g[f"{tag}/grad/{k}"]) < 1e-4, k
    for k, v in zip(g[f"{tag}/running_names"], g[f"{tag}/running_chk"]):
        t = sd[str(k)].double()
        np.testing.assert_allclose([t.sum().item(), t.abs().sum().item()], v, rtol=1e-5)
    with torch.no_grad():
        assert rel_err(R.resnet18_forward(sd, x, False).numpy(), g[f"{tag}/eval_out"]) < RTOL
