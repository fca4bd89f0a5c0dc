"""Host-side pieces either side of the hot path, no GPU needed: C-index, accumulate-to-64 rule, blender
update cadence, the blender's classification branch (host logic with a torch loss callable), deep-copy safety of the native modules."""
import copy
import itertools
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import main as cli  # noqa: E402
from tests._util import load_golden  # noqa: E402
from tests.test_oracle import _cls_inputs, _cls_update_inputs  # noqa: E402


def _brute_force_c(times, scores, events):
    """Pair counting written as lifelines documents it: process subjects by exit time, deaths before censored at equal times;
    each subject is compared with every death that happened strictly earlier (deaths) / earlier or at the same time (censored)."""
    order = sorted(range(len(times)), key=lambda i: (times[i], 0 if events[i] else 1))
    pairs = correct = tied = 0
    for pos, b in enumerate(order):
        for a in order[:pos]:
            if not events[a]:
                continue
            if times[a] == times[b] and events[b]:
                continue          # two deaths at the same time are not comparable
            pairs += 1
            correct += scores[a] < scores[b]
            tied += scores[a] == scores[b]
    return (correct + 0.5 * tied) / pairs if pairs else float("nan")


def test_concordance_index_vs_pair_count():
    rng = np.random.default_rng(5)
    for n in (2, 3, 7, 40):
        for _ in range(20):
            t = rng.integers(1, 6, n).astype(float)              # few distinct times: many ties
            s = np.round(rng.standard_normal(n), 1)               # score ties too
            e = (rng.random(n) < 0.6).astype(int)
            a, b = cli.concordance_index(t, s, e), _brute_force_c(list(t), list(s), list(e))
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-12, (t, s, e, a, b)
    # hand-checked: deaths at t=1,2,3 and a censored case at t=3 (tied with a death: admissible, score tie: half credit)
    assert abs(cli.concordance_index([1, 2, 3, 3], [0.1, 0.5, 0.2, 0.2], [1, 1, 1, 0]) - 3.5 / 6) < 1e-12
    assert cli.concordance_index([1, 2, 3], [1, 2, 3], [1, 1, 1]) == 1.0 and cli.concordance_index([1, 2, 3], [3, 2, 1], [1, 1, 1]) == 0.0
    assert np.isnan(cli.concordance_index([1, 2], [0, 1], [0, 0]))
    p = rng.standard_normal((9, 2)); ev = (rng.random((9, 2)) < 0.7).astype(int); du = rng.integers(1, 50, (9, 2))
    assert cli.getCIndices(p, ev, du) == [cli.concordance_index(du[:, i], p[:, i], ev[:, i]) for i in range(2)]


def test_accumulate_to_64_rule_and_cadence():
    """main.py:403-407,478-481: optimizer steps after every 64 / batch micro-batches and after the last one; :584: blender update."""
    assert cli.super_batch_interval(2) == 32 and cli.super_batch_interval(8) == 8 and cli.super_batch_interval(64) == 1
    assert cli.super_batch_interval(2, world=8) == 4 and cli.super_batch_interval(128) == 1
    for n_patients, bs in itertools.product((64, 100, 130, 7), (2, 8)):
        n_batches = -(-n_patients // bs)
        k = cli.super_batch_interval(bs)
        steps = [i for i in range(n_batches) if cli.is_step_boundary(i, n_batches, k)]
        assert steps[-1] == n_batches - 1
        assert len(steps) == cli.optimizer_steps_per_epoch(n_batches, k)
        if n_patients % bs == 0:
            assert len(steps) == -(-n_patients // 64)            # upstream's steps_per_epoch (:404-407)
        assert all(b - a == k for a, b in zip(steps[:-2], steps[1:-1]))
    assert [e for e in range(10) if cli.blender_update_due(e, 5)] == [4, 9]
    assert [e for e in range(4) if cli.blender_update_due(e, 1)] == [0, 1, 2, 3]


@pytest.mark.parametrize("red", ["sum", "mean"])
def test_blender_classification_branch_host_logic(red):
    """GradientBlender's classification branch with a torch loss callable on CPU tensors vs the reference (G9)."""
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    g = load_golden("g9_classification.npz")
    bce = torch.nn.BCEWithLogitsLoss(pos_weight=torch.from_numpy(g["pos_weight"]), reduction='none')
    logits, targets = _cls_inputs()
    x = logits.clone().requires_grad_(True)
    gb = GradientBlender(bce, reduction=red)
    loss = gb.computeLoss(x, targets)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{red}/loss"][0], rtol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g[f"{red}/grad"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gb.computeLoss(logits, targets, reduceToHeads=True).numpy(), g[f"{red}/heads"], rtol=1e-6)
    np.testing.assert_allclose(gb.computeLoss(logits, targets, no_reduce=True).numpy(), g["no_reduce"], rtol=1e-6)
    assert len(gb.history) == 1                                   # the first classification loss records the initial weights
    if red == "sum":
        gb = GradientBlender(bce)
        for it in range(3):
            gb.updateWeights(*_cls_update_inputs(it))
            np.testing.assert_allclose(gb.weights.numpy(), g["upd_weights"][it], rtol=2e-4, atol=1e-6)
            np.testing.assert_allclose(np.stack([gb.ltn.numpy(), gb.lvn.numpy()]), g["upd_losses"][it], rtol=1e-6)
        assert len(gb.history) == 1                               # ... and updateWeightsClass does not append (upstream :134-136)
    with pytest.raises(ValueError, match="Unable to reduce loss, unrecognized reduction: bogus"):
        GradientBlender(bce, reduction="bogus").computeLoss(logits, targets)


def test_backbone_copies_do_not_share_native_state():
    """copy.deepcopy / pickle of a model must not duplicate plan handles or flat-buffer views (double free)."""
    import pickle
    from mmnn_sts_amd.models.densenet import DenseNet
    m = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, block_config=(2, 2))
    bb = m.backbone
    bb.flat_parameters                                             # flatten: parameters become views of one buffer
    bb._plans[("fake", 0)] = {"plan": 12345, "ws": None}           # stands for a live native handle
    try:
        for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
            cb = clone.backbone
            assert len(cb._plans) == 0 and cb._flat is None and cb._params is None and cb._anchor is None
            assert cb.conv0.weight.data_ptr() != bb.conv0.weight.data_ptr()
            assert torch.equal(cb.conv0.weight, bb.conv0.weight) and torch.equal(cb.norm5.running_var, bb.norm5.running_var)
            assert list(clone.state_dict().keys()) == list(m.state_dict().keys())
            flat = cb.flat_parameters                              # the copy re-flattens lazily, on its own storage
            assert flat.data_ptr() != bb._flat.data_ptr() and cb.conv0.weight.data_ptr() == flat.data_ptr()
    finally:
        bb._plans.clear()                                          # the fake handle must not reach plan_destroy


def test_cli_flag_surface():
    """The reference's flags (main.py:898-947) parse, including the SageMaker string twins."""
    a = cli.build_arg_parser().parse_args(["--images", "--preop", "--survival", "--blend", "--blend_update_interval", "3", "--epochs", "4",
                                           "--lr", "1e-3", "--use_postop", "true", "--weights", "w.pth", "--config", "c.yaml"])
    assert a.images and a.preop and a.survival and a.blend and a.blend_update_interval == 3 and a.epochs == 4 and a.use_postop == "true"
    for flag in ("postop", "radiomics", "classification", "segmentation", "lr_finder", "no_gradcam", "inference", "split", "bootstrap"):
        assert getattr(cli.build_arg_parser().parse_args([f"--{flag}"]), flag) is True
    assert cli.str_to_bool("True") is True and cli.str_to_bool("false") is False
    with pytest.raises(ValueError):
        cli.str_to_bool("maybe")


def test_synthetic_csv_round_trip(tmp_path):
    preds = [f"predictor{i}" for i in range(32)]
    path = cli.write_synthetic_csv(str(tmp_path / "p.csv"), 64, preds, seed=3)
    ds = cli.ClinicalCsvDataset(path, preds)
    assert len(ds) == 64 and ds.clinical.shape == (64, 32) and ds.events.shape == (64, 2) and ds.durations.dtype == torch.int64
    x, ev, du = cli.collate([ds[0], ds[1]])
    assert x.shape == (2, 32) and x.dtype == torch.float32 and ev.shape == (2, 2)
    with pytest.raises(ValueError):
        cli.ClinicalCsvDataset(path, preds + ["missing_column"])


def test_bootstrap_c_indices():
    """`--bootstrap` (main.py:767-768,857-887): 50 resamples with replacement of the evaluated patients, mean / std of the C-indices;
    resamples without an admissible pair are skipped."""
    rng = np.random.default_rng(11)
    n = 24
    du = rng.integers(1, 500, (n, 2)); ev = (rng.random((n, 2)) < 0.7).astype(int)
    p = du / 500.0 + 0.05 * rng.standard_normal((n, 2))           # informative scores (lifelines' convention: a higher score = outlives)
    means, stds, used = cli.bootstrap_c_indices(p, ev, du, iterations=50, seed=3)
    assert used == 50 and len(means) == len(stds) == 2
    full = cli.getCIndices(p, ev, du)
    for m, s_, f in zip(means, stds, full):
        assert 0.0 < s_ < 0.2 and abs(m - f) < 3 * s_ + 0.05 and m > 0.7
    again = cli.bootstrap_c_indices(p, ev, du, iterations=50, seed=3)
    assert again[0] == means and again[1] == stds                  # deterministic for a seed
    # no admissible pair in any resample (nobody died): nothing usable, NaN summary, no exception
    m0, s0, u0 = cli.bootstrap_c_indices(p, np.zeros_like(ev), du, iterations=5)
    assert u0 == 0 and all(np.isnan(m0)) and all(np.isnan(s0))
    # --bootstrap outside `--inference --survival` is rejected loudly
    with pytest.raises(SystemExit):
        cli.main(["--images", "--survival", "--bootstrap"])


def test_load_weights_bhb_remap(tmp_path):
    """utils/utils.py:357-390: a BHB-10K style checkpoint ({'model': {'module.features.denseblockB.denselayerL.<leaf>': ...}}) gets
    'module.' stripped and 'layers' inserted after the dense layer; `strict=False`.  Quirk Q13: the remapped keys are
    `features.*` while this DenseNet registers its backbone as `backbone.*`, so NONE of the convolutional weights land -- reproduced;
    keys that do exist in the model (here: the classifier) are loaded."""
    from mmnn_sts_amd.models.densenet import DenseNet
    from mmnn_sts_amd.utils.utils import loadWeights, remap_bhb_keys
    torch.manual_seed(0)
    m = DenseNet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, block_config=(2, 2))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ck = {"model": {
        "module.features.conv0.weight": torch.full((64, 1, 7, 7, 7), 7.0),
        "module.features.denseblock1.denselayer1.norm1.weight": torch.full((64,), 3.0),
        "module.features.denseblock1.denselayer2.conv2.weight": torch.full((32, 128, 3, 3, 3), 5.0),
        "module.features.transition1.conv.weight": torch.full((64, 128, 1, 1, 1), 9.0),
        "module.class_layers.out.weight": torch.full((2, 12), 0.25),
    }}
    remapped = remap_bhb_keys(ck["model"])
    assert set(remapped) == {"features.conv0.weight", "features.denseblock1.denselayer1.layers.norm1.weight",
                             "features.denseblock1.denselayer2.layers.conv2.weight", "features.transition1.conv.weight",
                             "class_layers.out.weight"}
    path = str(tmp_path / "DenseNet121_BHB-10K_yAwareContrastive.pth")
    torch.save(ck, path)
    loadWeights(m, path, "cpu")
    after = m.state_dict()
    assert torch.equal(after["class_layers.out.weight"], torch.full((2, 12), 0.25))               # the one key that exists lands
    for k, v in before.items():
        if k != "class_layers.out.weight":
            assert torch.equal(after[k], v), k                                                     # Q13: the backbone is untouched
    # a plain state_dict under any other file name loads strictly
    plain = str(tmp_path / "model.pth")
    sd = {k: v + 1 if v.is_floating_point() else v for k, v in before.items()}
    torch.save(sd, plain)
    loadWeights(m, plain, "cpu")
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError):
        torch.save({"unexpected.key": torch.zeros(1)}, plain)
        loadWeights(m, plain, "cpu")
