"""Generate tests/golden/*.npz by running the REFERENCE's own classes (BUILD CONTAINER ONLY).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [--out tests/golden]

TEST INFRASTRUCTURE ONLY.  Reads /root/reference through oracle/ref_shim.py; never runs on the GPU box and is never
imported by tests/, smoke() or bench.py.  Fixtures contain numbers only (inputs are re-derived from the
oracle/synth.py formula, so only expected outputs are stored).  Dropout probabilities are forced to 0 because RNG
streams cannot be matched bit-wise; everything else is the reference's arithmetic as published.

Fixture families: G1 mlp, G2 densenet, G3 fusion train step, G4 fusion eval, G5 gradcam,
G6 blender update sequence + Cox known answers (+ fractional durations), G7 tiny densenet, G8 unimodal (BASELINE config 2)
training step, G9 classification path of the blender / pos-weighted BCE, G10 r3d_18.  G3 / G5 / G8 also exist at the BASELINE
extents (128^3, Grad-CAM 256^3): `--only g3big,g5big,g8` (minutes of CPU time, kept out of the default set).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, synth  # noqa: E402

ref_shim.install()
from losses.GradientBlender import GradientBlender  # noqa: E402  (reference)
from losses.losses import CoxPH  # noqa: E402  (reference)
from models.densenet import DenseNet121, TinyDensenet  # noqa: E402  (reference)
from models.mlp import MLP  # noqa: E402  (reference)
from models.multimodal import MultiModalModel  # noqa: E402  (reference)
from utils.utils import MultiModalGradCAM, surv_criterion  # noqa: E402  (reference)

N_CLIN = 32


def load_synth(model: torch.nn.Module, prefix: str = "") -> None:
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(shapes, prefix).items()}
    model.load_state_dict(sd, strict=True)


def zero_dropout(model: torch.nn.Module) -> None:
    for m in model.modules():
        if m.__class__.__name__.startswith("Dropout"):
            m.p = 0.0


def labels(n: int):
    if n == 2:
        ev = np.array([[1, 0], [0, 1]], dtype=np.int64)
        du = np.array([[100, 250], [300, 50]], dtype=np.int64)
    else:
        ev = (synth.uniform(f"events/{n}", (n, 2)) > 0).astype(np.int64)
        ev[0, :] = 1
        du = (1 + np.floor((synth.uniform(f"durations/{n}", (n, 2)) * 0.5 + 0.5) * 2998)).astype(np.int64)
    return torch.from_numpy(ev), torch.from_numpy(du)


def image_in(n, c, s):
    return torch.from_numpy(synth.uniform(f"image/{n}x{c}x{s}", (n, c, s, s, s)))


def clin_in(n):
    return torch.from_numpy(synth.uniform(f"clinical/{n}", (n, N_CLIN)))


def stat3(t: torch.Tensor):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item()], dtype=np.float64)


def grad_probes(model: torch.nn.Module):
    names, norms, sums, heads = [], [], [], []
    for k, p in model.named_parameters():
        names.append(k)
        if p.grad is None:
            norms.append(np.nan), sums.append(np.nan), heads.append(np.full(8, np.nan, np.float32))
            continue
        g = p.grad.detach()
        norms.append(g.double().norm().item())
        sums.append(g.double().sum().item())
        h = np.zeros(8, np.float32)
        flat = g.flatten()[:8].numpy()
        h[: flat.size] = flat
        heads.append(h)
    return {"grad_names": np.array(names), "grad_l2": np.array(norms), "grad_sum": np.array(sums),
            "grad_head": np.stack(heads)}


def bn_running(model: torch.nn.Module):
    """(sum, abs-sum) of every running_mean / running_var + num_batches_tracked, in state_dict order."""
    names, vals = [], []
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            names.append(k)
            vals.append([v.double().sum().item(), v.double().abs().sum().item()])
    return {"running_names": np.array(names), "running_chk": np.array(vals)}


def g1_mlp(out):
    res = {}
    for n in (2, 8):
        m = MLP(N_CLIN, 2, 12)
        load_synth(m, "mlp.")
        zero_dropout(m)
        x = clin_in(n)
        m.eval()
        with torch.no_grad():
            res[f"eval_out_n{n}"] = m(x).numpy()
        m.train()
        f = m.features(m.backbone(x))
        y = m.output_head(f)
        (y * torch.from_numpy(synth.uniform("mlp/cot", tuple(y.shape)))).sum().backward()
        res[f"train_feat_n{n}"] = f.detach().numpy()
        res[f"train_out_n{n}"] = y.detach().numpy()
        for k, p in m.named_parameters():
            res[f"grad_n{n}/{k}"] = p.grad.numpy()
        for k, v in m.state_dict().items():
            if "running" in k:
                res[f"run_n{n}/{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(out, "g1_mlp.npz"), **res)


def _densenet_case(cls, in_ch, s, n=2):
    m = cls(spatial_dims=3, in_channels=in_ch, out_channels=2, feature_channels=12, dropout_prob=0.2)
    load_synth(m, "densenet.")
    zero_dropout(m)
    m.train()
    taps = {}
    hooks = []
    for name, mod in m.backbone.named_children():
        if name in ("conv0", "pool0", "norm5") or name.startswith("denseblock") or name.startswith("transition"):
            hooks.append(mod.register_forward_hook(lambda _m, _i, o, nm=name: taps.__setitem__(nm, stat3(o))))
    x = image_in(n, in_ch, s)
    h = m.backbone(x)
    h_pre = h.detach().clone()          # features.relu is in-place (models/densenet.py:237): copy before it runs
    f = m.features(h)
    y = m.class_layers(f)
    for hk in hooks:
        hk.remove()
    res = {"norm5": h_pre.numpy(), "features": f.detach().numpy(), "out": y.detach().numpy()}
    res["tap_names"] = np.array(list(taps.keys()))
    res["tap_stats"] = np.stack([taps[k] for k in taps])
    res.update(bn_running(m))
    sd = m.state_dict()
    for k in ("backbone.norm0.running_mean", "backbone.norm0.running_var", "backbone.norm5.running_mean",
              "backbone.norm5.running_var", "backbone.denseblock1.denselayer1.layers.norm2.running_var"):
        res["run/" + k] = sd[k].numpy().copy()
    res["nbt"] = np.array([sd["backbone.norm0.num_batches_tracked"].item()])
    # eval-mode forward with the (now updated) running statistics
    m.eval()
    with torch.no_grad():
        res["eval_out"] = m(x).numpy()
    return res


def g2_densenet(out):
    for in_ch, s in ((1, 32), (2, 32), (2, 64)):
        np.savez_compressed(os.path.join(out, f"g2_densenet_in{in_ch}_s{s}.npz"), **_densenet_case(DenseNet121, in_ch, s))


def g7_tiny(out):
    np.savez_compressed(os.path.join(out, "g7_tiny_in2_s32.npz"), **_densenet_case(TinyDensenet, 2, 32))


def build_fusion(blend: bool):
    img = DenseNet121(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=0.2)
    mm = MultiModalModel(img, [f"p{i}" for i in range(N_CLIN)], 2, 12, blend=blend)
    load_synth(mm, "fusion.")
    zero_dropout(mm)
    return mm


def g3_g4_fusion(out, sizes=(32, 64)):
    for s in sizes:
        res = {}
        for blend in (True, False):
            tag = "blend" if blend else "plain"
            mm = build_fusion(blend)
            mm.train()
            n = 2
            x = {"image": image_in(n, 2, s), "clinical": clin_in(n)}
            ev, du = labels(n)
            outp = mm(x)
            if blend:
                gb = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
                loss, sel = gb.computeLoss(outp, ev, du)
                res[f"{tag}/head_losses"] = gb.computeLossSurv(outp, ev, du, reduceToHeads=True).detach().numpy()
                res[f"{tag}/weights"] = gb.weights.numpy()
                res[f"{tag}/selection_loss"] = np.array([sel.item()])
            else:
                loss = surv_criterion(CoxPH, outp, ev, du, "cpu")
            loss.backward()
            res[f"{tag}/out"] = outp.detach().numpy()
            res[f"{tag}/loss"] = np.array([loss.item()], dtype=np.float64)
            for k, v in grad_probes(mm).items():
                res[f"{tag}/{k}"] = v
            gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in mm.parameters() if p.grad is not None)).item()
            res[f"{tag}/grad_global_l2"] = np.array([gn])
            for k in ("output_head.weight", "image_output_head.weight", "clinical_output_head.weight",
                      "image_model.model.features.feature_layer.bias", "clinical_model.model.backbone.dense0.weight",
                      "image_model.model.backbone.norm5.weight", "image_model.model.backbone.norm0.weight",
                      "image_model.model.backbone.norm0.bias",
                      "image_model.model.backbone.denseblock1.denselayer1.layers.norm1.weight",
                      "image_model.model.backbone.denseblock4.denselayer16.layers.conv2.weight"):
                p = dict(mm.named_parameters())[k]
                if p.grad is not None and (s < 128 or p.grad.numel() <= 4096):
                    res[f"{tag}/grad/{k}"] = p.grad.numpy()
            if s == 32:
                p = dict(mm.named_parameters())["image_model.model.backbone.conv0.weight"]
                res[f"{tag}/grad/image_model.model.backbone.conv0.weight"] = p.grad.numpy()
            for k, v in bn_running(mm).items():
                res[f"{tag}/{k}"] = v
            # G4: eval-mode forward after the one training step's running-stat update
            mm.eval()
            with torch.no_grad():
                res[f"{tag}/eval_out"] = mm(x).numpy()
        np.savez_compressed(os.path.join(out, f"g3_fusion_s{s}.npz"), **res)


def g5_gradcam(out, s=64):
    mm = build_fusion(False)
    mm.eval()
    cam = MultiModalGradCAM(mm)
    x = {"image": image_in(1, 2, s), "clinical": clin_in(1)}
    preds, maps = cam(x)
    res = {"preds": preds.detach().numpy()}
    for i, m in enumerate(maps):
        m = m.detach()
        res[f"map{i}_coarse"] = m[:: s // 8, :: s // 8, :: s // 8].numpy()  # 8^3 sub-sample of the upsampled map
        res[f"map{i}_stats"] = stat3(m)
        res[f"map{i}_corner"] = m[:4, :4, :4].numpy()
    res["act_after"] = cam.features.detach().numpy()      # activations after the cumulative in-place weighting
    res["last_grads"] = cam.grads.detach().numpy()         # gradient of class C-1 at the hooked conv
    np.savez_compressed(os.path.join(out, f"g5_gradcam_s{s}.npz"), **res)


def g8_unimodal(out, sizes=(64, 128)):
    """BASELINE config 2 (`--images --survival`, modality t1): DenseNet121(in=1) full forward incl. class_layers ->
    surv_criterion(CoxPH) -> backward  (reference main.py:451,460,466,469)."""
    for s in sizes:
        m = DenseNet121(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, dropout_prob=0.2)
        load_synth(m, "densenet.")
        zero_dropout(m)
        m.train()
        n = 2
        x = image_in(n, 1, s)
        ev, du = labels(n)
        y = m(x)
        loss = surv_criterion(CoxPH, y, ev, du, "cpu")
        loss.backward()
        res = {"out": y.detach().numpy(), "loss": np.array([loss.item()], dtype=np.float64)}
        res.update(grad_probes(m))
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
        res["grad_global_l2"] = np.array([gn])
        for k in ("class_layers.out.weight", "class_layers.out.bias", "features.feature_layer.weight", "backbone.norm5.weight",
                  "backbone.norm0.weight", "backbone.denseblock1.denselayer1.layers.norm1.weight"):
            res[f"grad/{k}"] = dict(m.named_parameters())[k].grad.numpy()
        res.update(bn_running(m))
        m.eval()
        with torch.no_grad():
            res["eval_out"] = m(x).numpy()
        np.savez_compressed(os.path.join(out, f"g8_unimodal_in1_s{s}.npz"), **res)


def g6_blender(out):
    res = {}
    # known answers of SURVEY 8(c)
    res["kat1"] = np.array([CoxPH(torch.tensor([.3, -.2, .1, .4]), torch.tensor([1, 0, 1, 1]),
                                  torch.tensor([100, 250, 300, 50])).item()])
    P = torch.tensor([[.3, -.2], [.1, .4], [-.5, .2], [0, .7]])
    E = torch.tensor([[1, 0], [0, 1], [1, 1], [0, 0]])
    D = torch.tensor([[100, 250], [300, 50], [20, 400], [75, 75]])
    res["kat2"] = np.array([surv_criterion(CoxPH, P, E, D, "cpu").item()])
    preds = torch.tensor([[[.3, -.2], [.1, .4]], [[.5, 0], [-.1, .2]], [[0, .1], [.2, -.3]]])
    ev, du = labels(2)
    gb = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
    loss, sel = gb.computeLoss(preds, ev, du)
    res["kat3_loss"] = np.array([loss.item()])
    res["kat3_sel"] = np.array([sel.item()])
    res["kat3_heads"] = gb.computeLossSurv(preds, ev, du, reduceToHeads=True).numpy()
    # update sequence, L = 12 train / 10 val patients (<= 16: tie order of torch.sort is the stable one)
    gb = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
    hist_w, hist_l = [], []
    for it in range(3):
        tp = torch.from_numpy(synth.uniform(f"gb/train/{it}", (3, 12, 2)))
        vp = torch.from_numpy(synth.uniform(f"gb/val/{it}", (3, 10, 2)))
        te = torch.from_numpy((synth.uniform("gb/te", (12, 2)) > -0.2).astype(np.int64))
        ve = torch.from_numpy((synth.uniform("gb/ve", (10, 2)) > -0.2).astype(np.int64))
        td = torch.from_numpy((1 + np.floor((synth.uniform("gb/td", (12, 2)) * .5 + .5) * 2998)).astype(np.int64))
        vd = torch.from_numpy((1 + np.floor((synth.uniform("gb/vd", (10, 2)) * .5 + .5) * 2998)).astype(np.int64))
        gb.updateWeights(tp, te, td, vp, ve, vd)
        hist_w.append(gb.weights.numpy().copy())
        hist_l.append(np.stack([gb.ltn.numpy(), gb.lvn.numpy()]))
    res["upd_weights"] = np.stack(hist_w)
    res["upd_losses"] = np.stack(hist_l)
    res["upd_history"] = np.array(gb.history)
    np.savez_compressed(os.path.join(out, "g6_blender.npz"), **res)


def g9_classification(out):
    """Classification branch of the blender + the pos-weighted BCE of the classification trainer (reference main.py:147-156,
    207-212, 264, 311-314; losses/GradientBlender.py:105-136,150-179; utils/utils.py:20-22), and the Cox loss on FRACTIONAL
    (float32) durations / events as the reference's datasets build them (data/ImageDatasets.py:462)."""
    from utils.utils import criterion  # noqa: E402  (reference)
    res = {}
    freqs = torch.tensor([0.3, 0.45])
    pw = (torch.ones_like(freqs) - freqs) / freqs
    res["pos_weight"] = pw.numpy()
    bce_sum = torch.nn.BCEWithLogitsLoss(pos_weight=pw, reduction='sum')
    bce_none = torch.nn.BCEWithLogitsLoss(pos_weight=pw, reduction='none')
    n = 6
    logits = torch.from_numpy(synth.uniform("cls/logits", (3, n, 2), 2.0)).requires_grad_(True)
    targets = torch.from_numpy((synth.uniform("cls/targets", (n, 2)) > 0).astype(np.float32))
    res["criterion_sum"] = np.array([criterion(bce_sum, logits[0], targets, "cpu").item()])
    for red in ("sum", "mean"):
        gb = GradientBlender(bce_none, reduction=red, device="cpu")
        loss = gb.computeLoss(logits, targets)
        g, = torch.autograd.grad(loss, logits)
        res[f"{red}/loss"] = np.array([loss.item()])
        res[f"{red}/grad"] = g.numpy()
        res[f"{red}/heads"] = gb.computeLoss(logits, targets, reduceToHeads=True).detach().numpy()
        res[f"{red}/history_len"] = np.array([len(gb.history)])
    gb = GradientBlender(bce_none, device="cpu")
    res["no_reduce"] = gb.computeLoss(logits, targets, no_reduce=True).detach().numpy()
    # update sequence as train_classification feeds it: sigmoid probabilities for train, thresholded predictions for val
    gb = GradientBlender(bce_none, device="cpu")
    ws, ls = [], []
    for it in range(3):
        tp = torch.sigmoid(torch.from_numpy(synth.uniform(f"cls/train/{it}", (3, 12, 2), 2.0)))
        vp = (torch.sigmoid(torch.from_numpy(synth.uniform(f"cls/val/{it}", (3, 10, 2), 2.0))) > 0.5).float()
        tt = torch.from_numpy((synth.uniform("cls/tt", (12, 2)) > 0).astype(np.float32))
        vt = torch.from_numpy((synth.uniform("cls/vt", (10, 2)) > 0).astype(np.float32))
        gb.updateWeights(tp, tt, vp, vt)
        ws.append(gb.weights.numpy().copy())
        ls.append(np.stack([gb.ltn.numpy(), gb.lvn.numpy()]))
    res["upd_weights"], res["upd_losses"], res["upd_history_len"] = np.stack(ws), np.stack(ls), np.array([len(gb.history)])
    # Cox loss with fractional durations (float32 tensors, as torch.Tensor([...]) builds them upstream)
    for n in (4, 9):
        h = torch.from_numpy(synth.uniform(f"coxf/h{n}", (n, 2)))
        ev = torch.from_numpy((synth.uniform(f"coxf/e{n}", (n, 2)) > -0.3).astype(np.float32))
        ev[0] = 1
        du = torch.from_numpy((synth.uniform(f"coxf/d{n}", (n, 2)) * 0.5 + 0.5).astype(np.float32) * 30.0 + 0.25)
        res[f"coxf/n{n}"] = np.array([surv_criterion(CoxPH, h, ev, du, "cpu").item()])
        res[f"coxf/n{n}/c0"] = np.array([CoxPH(h[:, 0], ev[:, 0], du[:, 0]).item()])
    np.savez_compressed(os.path.join(out, "g9_classification.npz"), **res)


def g10_r3d18(out):
    """r3d_18 (reference models/resnet.py:202-227): training forward (dropout forced to 0) + backward of sum(out * cot), running
    statistics, eval forward."""
    from models.resnet import r3d_18  # noqa: E402  (reference)
    res = {}
    for tag, shape in (("a", (2, 1, 16, 64, 64)), ("b", (3, 1, 9, 40, 52))):
        m = r3d_18(2)
        load_synth(m, "r3d.")
        zero_dropout(m)
        m.train()
        x = torch.from_numpy(synth.uniform(f"r3d/x/{tag}", shape))
        y = m(x)
        cot = torch.from_numpy(synth.uniform(f"r3d/cot/{tag}", tuple(y.shape)))
        (y * cot).sum().backward()
        res[f"{tag}/out"] = y.detach().numpy()
        for k, v in grad_probes(m).items():
            res[f"{tag}/{k}"] = v
        for k in ("fc.weight", "fc.bias", "stem.0.weight", "stem.1.weight", "layer1.0.downsample.0.weight", "layer4.1.conv2.0.weight",
                  "layer2.0.conv1.0.weight", "layer3.0.downsample.1.bias"):
            res[f"{tag}/grad/{k}"] = dict(m.named_parameters())[k].grad.numpy()
        for k, v in bn_running(m).items():
            res[f"{tag}/{k}"] = v
        m.eval()
        with torch.no_grad():
            res[f"{tag}/eval_out"] = m(x).numpy()
    np.savez_compressed(os.path.join(out, "g10_r3d18.npz"), **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    jobs = {"g1": g1_mlp, "g2": g2_densenet, "g3": g3_g4_fusion, "g5": g5_gradcam, "g6": g6_blender, "g7": g7_tiny,
            "g8": g8_unimodal, "g9": g9_classification, "g10": g10_r3d18}
    big = {"g3big": lambda o: g3_g4_fusion(o, sizes=(128,)), "g5big": lambda o: [g5_gradcam(o, 128), g5_gradcam(o, 256)]}
    jobs.update(big)
    for k, fn in jobs.items():
        if (a.only and k not in a.only.split(",")) or (not a.only and k in big):
            continue
        print("generating", k, flush=True)
        fn(a.out)
    print("done ->", a.out)


if __name__ == "__main__":
    main()
