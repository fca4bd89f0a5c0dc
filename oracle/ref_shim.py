"""Import shim that lets the reference's OWN model / loss classes run in the build container.

TEST INFRASTRUCTURE ONLY, BUILD CONTAINER ONLY: it reads /root/reference at run time, which does not exist on the
GPU box; nothing in tests/ (gpu or not), smoke() or bench.py imports this module.  It is used by
oracle/make_golden.py to produce the numeric fixtures under tests/golden/.

The reference imports monai / pycox / torchvision / medcam / boto3 / ... which are not installed.  The six monai
symbols its models use are pure look-ups of torch.nn classes (models/densenet.py:24-26,71-85,142-148,190-202), so
stubbing them leaves the arithmetic to PyTorch unchanged.  pycox's CoxPHLoss is restated (see restatement.py).
Everything else is inert.
"""
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _mod(name: str) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave as a package

        def _inert(attr, _n=name):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return type(attr, (), {"__init__": lambda self, *a, **k: None})

        m.__getattr__ = _inert  # any other symbol the reference imports resolves to an inert class
        sys.modules[name] = m
        if "." in name:
            parent, leaf = name.rsplit(".", 1)
            setattr(_mod(parent), leaf, m)
    return m


class _Factory:
    def __init__(self, table):
        self._t = table

    def __getitem__(self, key):
        name, dim = key
        return self._t[name.lower()][int(dim)]


def install() -> None:
    if getattr(install, "_done", False):
        return
    fac = _mod("monai.networks.layers.factories")

    class Conv(_Factory):
        CONV = "conv"

    class Dropout(_Factory):
        DROPOUT = "dropout"

    class Pool(_Factory):
        MAX, AVG, ADAPTIVEAVG = "max", "avg", "adaptiveavg"

    fac.Conv = Conv({"conv": {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}})
    fac.Dropout = Dropout({"dropout": {1: nn.Dropout, 2: nn.Dropout2d, 3: nn.Dropout3d}})
    fac.Pool = Pool({
        "max": {1: nn.MaxPool1d, 2: nn.MaxPool2d, 3: nn.MaxPool3d},
        "avg": {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d},
        "adaptiveavg": {1: nn.AdaptiveAvgPool1d, 2: nn.AdaptiveAvgPool2d, 3: nn.AdaptiveAvgPool3d},
    })
    for n in ("CONV", "DROPOUT", "MAX", "AVG", "ADAPTIVEAVG"):
        for obj in (fac.Conv, fac.Dropout, fac.Pool):
            if hasattr(type(obj), n):
                setattr(obj, n, getattr(type(obj), n))

    lu = _mod("monai.networks.layers.utils")

    def get_norm_layer(name, spatial_dims=1, channels=1):
        assert (name if isinstance(name, str) else name[0]).lower() == "batch"
        return {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[spatial_dims](channels)

    def get_act_layer(name):
        kind, kw = (name, {}) if isinstance(name, str) else name
        assert kind.lower() == "relu"
        return nn.ReLU(**kw)

    lu.get_norm_layer, lu.get_act_layer = get_norm_layer, get_act_layer
    _mod("monai.utils.module").look_up_option = lambda opt, table, default=None: table.get(opt, default)
    _mod("monai.utils.type_conversion").convert_to_tensor = torch.as_tensor
    nets = _mod("monai.networks.nets")
    for n in ("densenet121", "DenseNet121", "Densenet201", "SEResNet50"):
        setattr(nets, n, type(n, (), {}))
    tr = _mod("monai.transforms")
    tr.Transform = type("Transform", (), {})
    for n in ("Compose", "RandRotate", "RandFlip", "RandZoom", "Resize", "ScaleIntensity", "EnsureType"):
        setattr(tr, n, type(n, (), {}))
    for n in ("monai.config", "monai.metrics", "monai.losses", "monai.data"):
        _mod(n)

    # pycox: restated (not installed, not vendored) -- keep the (log_h, durations, events) signature.
    from oracle.restatement import pycox_cox_ph_loss

    class CoxPHLoss(nn.Module):
        def forward(self, log_h, durations, events):
            return pycox_cox_ph_loss(log_h, durations, events)

    _mod("pycox.models.loss").CoxPHLoss = CoxPHLoss

    # inert stubs
    _mod("torchvision")
    med = _mod("medcam")
    med.medcam = types.SimpleNamespace(inject=lambda *a, **k: None)
    _mod("boto3")
    _mod("botocore.exceptions")
    _mod("SimpleITK")
    _mod("nibabel")
    _mod("skmultilearn.model_selection.iterative_stratification").iterative_train_test_split = None
    _mod("skmultilearn.model_selection").iterative_train_test_split = None
    _mod("lifelines.utils").concordance_index = None
    _mod("torch_lr_finder")
    try:
        import matplotlib  # noqa: F401
    except Exception:
        _mod("matplotlib.pyplot")

    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True
