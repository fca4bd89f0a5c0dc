"""MI355X-native mirror of the reference's 3-D DenseNet (models/densenet.py:151-356 of DigITs-AIML/MMNN_STS).

Same constructor signature, same attribute names (`backbone`, `features`, `class_layers`) and the same `state_dict`
schema as the reference, so checkpoints and callers (`utils.BackpropagatableFeatureExtractor`, `MultiModalModel`,
`parser.getModel`) are interchangeable -- but `backbone.forward` / `features.forward` run hand-written HIP kernels
through the C-ABI (include/mmnn_sts.h).  The torch.nn sub-modules below are PARAMETER CONTAINERS ONLY: their own
`forward` is never called; all their parameters are views into one flat fp32 buffer the kernels index directly.
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict
from typing import Sequence, Union

import torch
import torch.nn as nn

from .. import _lib, ops

__all__ = ["DenseNet", "DenseNet121", "TinyDensenet"]


class _DenseLayer(nn.Module):
    """Container for norm1/relu1/conv1/norm2/relu2/conv2[/dropout] (reference `_DenseLayer`, models/densenet.py:46-89)."""

    def __init__(self, in_channels: int, growth_rate: int, bn_size: int, dropout_prob: float):
        super().__init__()
        mid = bn_size * growth_rate
        self.layers = nn.Sequential()
        self.layers.add_module("norm1", nn.BatchNorm3d(in_channels))
        self.layers.add_module("relu1", nn.ReLU(inplace=True))
        self.layers.add_module("conv1", nn.Conv3d(in_channels, mid, kernel_size=1, bias=False))
        self.layers.add_module("norm2", nn.BatchNorm3d(mid))
        self.layers.add_module("relu2", nn.ReLU(inplace=True))
        self.layers.add_module("conv2", nn.Conv3d(mid, growth_rate, kernel_size=3, padding=1, bias=False))
        if dropout_prob > 0:
            self.layers.add_module("dropout", nn.Dropout3d(dropout_prob))


class _Backbone(nn.Sequential):
    """conv0 .. norm5 (models/densenet.py:196-231).  forward(x) = one C-ABI call into the HIP launch plan."""

    _MAX_PLANS = 3

    def __init__(self, in_channels, init_features, growth_rate, block_config, bn_size, dropout_prob):
        super().__init__()
        self.cfg = dict(in_channels=in_channels, init_features=init_features, growth_rate=growth_rate,
                        block_config=tuple(block_config), bn_size=bn_size, dropout_prob=float(dropout_prob))
        self.add_module("conv0", nn.Conv3d(in_channels, init_features, kernel_size=7, stride=2, padding=3, bias=False))
        self.add_module("norm0", nn.BatchNorm3d(init_features))
        self.add_module("relu0", nn.ReLU(inplace=True))
        self.add_module("pool0", nn.MaxPool3d(kernel_size=3, stride=2, padding=1))
        c = init_features
        for i, n_layers in enumerate(block_config):
            block = nn.Sequential()
            for j in range(n_layers):
                block.add_module(f"denselayer{j + 1}", _DenseLayer(c, growth_rate, bn_size, dropout_prob))
                c += growth_rate
            self.add_module(f"denseblock{i + 1}", block)
            if i == len(block_config) - 1:
                self.add_module("norm5", nn.BatchNorm3d(c))
            else:
                trans = nn.Sequential()
                trans.add_module("norm", nn.BatchNorm3d(c))
                trans.add_module("relu", nn.ReLU(inplace=True))
                trans.add_module("conv", nn.Conv3d(c, c // 2, kernel_size=1, bias=False))
                trans.add_module("pool", nn.AvgPool3d(kernel_size=2, stride=2))
                self.add_module(f"transition{i + 1}", trans)
                c //= 2
        self.out_channels = c
        self._reset_native_state()

    # ---- native state: flat buffers + launch plans.  Owned by exactly ONE module object -------------------------------
    _NATIVE_STATE = ("_flat", "_flat_run", "_flat_nbt", "_flat_grad", "_plans", "_anchor", "_params", "_bns", "_fwd_token",
                     "_grads_attached", "_grads_stale", "_explicit_version", "_grad_ready_hook")

    #: "always": the weights are re-packed for the kernels on every forward (safe with any way of writing parameters).
    #: "versioned": re-pack only when a parameter changed, as told by the autograd version counters of the parameters (every
    #: in-place op on a Parameter, `load_state_dict`, torch optimizers) plus `mark_params_changed()` (called by FusedSGD and by
    #: re-flattening).  Writes through `p.data` are invisible to version counters: call `mark_params_changed()` after them.
    repack_policy = "always"

    def _reset_native_state(self) -> None:
        """Non-module state (kept out of state_dict): nothing flattened, no plan.  Rebuilt lazily by the next forward."""
        object.__setattr__(self, "_flat", None)
        object.__setattr__(self, "_flat_run", None)
        object.__setattr__(self, "_flat_nbt", None)
        object.__setattr__(self, "_flat_grad", None)
        object.__setattr__(self, "_plans", OrderedDict())
        object.__setattr__(self, "_anchor", None)
        object.__setattr__(self, "_params", None)
        object.__setattr__(self, "_bns", None)
        object.__setattr__(self, "_fwd_token", 0)
        object.__setattr__(self, "_grads_attached", False)
        object.__setattr__(self, "_grads_stale", True)
        object.__setattr__(self, "_explicit_version", 1)
        object.__setattr__(self, "_grad_ready_hook", None)

    def set_grad_ready_hook(self, hook) -> None:
        """`hook(backbone, begin, end)` (or None): called from inside backward(), in stream order, as soon as the gradients
        `flat_grad[begin:end]` of this backward are final -- block 4 + norm5 first, then block 3 + its transition, ..., finally block 1 +
        transition 1 + stem.  With a hook installed the backward runs one C-ABI call per dense block
        (`mmnn_densenet_backward_range`), so a data-parallel caller can start the all-reduce of a range while the kernels of the
        next block run (mmnn_sts_amd/distributed.py: OverlappedGradientReducer).  Same kernels, same order, same results."""
        object.__setattr__(self, "_grad_ready_hook", hook)

    def block_param_ranges(self, ent):
        """[(begin, end)] of the flat buffers in backward completion order: last block first; the first block's range includes the stem."""
        r = ent.get("ranges")
        if r is None:
            L = _lib.lib()
            nb = len(self.cfg["block_config"])
            b, e = ctypes.c_int64(), ctypes.c_int64()
            r = []
            for blk in range(nb - 1, -1, -1):
                _lib.check(L.mmnn_densenet_block_param_range(ent["plan"], blk, ctypes.byref(b), ctypes.byref(e)), "block_param_range")
                r.append([blk, b.value, e.value])
            _lib.check(L.mmnn_densenet_block_param_range(ent["plan"], -1, ctypes.byref(b), ctypes.byref(e)), "block_param_range")
            assert e.value == r[-1][1] and b.value == 0
            r[-1][1] = 0                                   # the call for block 0 also finishes the stem
            r = ent["ranges"] = [tuple(t) for t in r]
        return r

    def mark_params_changed(self) -> None:
        """Tell the backbone that parameter values were written behind autograd's back (raw pointers, `p.data`)."""
        object.__setattr__(self, "_explicit_version", self._explicit_version + 1)

    def _params_version(self) -> int:
        v = self._explicit_version * 1000003 + self._flat._version * 7919
        for p in self._params:
            v += p._version
        return (v % ((1 << 62) - 1)) + 1          # non-zero

    def __getstate__(self):
        """copy.deepcopy(model) / torch.save(model) copy the module WITHOUT its native plan handles (raw pointers: two owners
        would free them twice) and without the flat views; the copy re-flattens and re-plans on its first forward."""
        state = super().__getstate__() if hasattr(nn.Module, "__getstate__") else self.__dict__.copy()
        state = dict(state)
        for k in self._NATIVE_STATE:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._reset_native_state()

    # ---- flat parameter storage -------------------------------------------------------------------------------------
    def _bn_modules(self):
        return [m for m in self.modules() if isinstance(m, nn.BatchNorm3d)]

    def _flatten(self) -> None:
        params = list(self.parameters())
        dev = params[0].device
        flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params])
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        bns = self._bn_modules()
        run = torch.cat([t.detach().reshape(-1).to(torch.float32) for m in bns for t in (m.running_mean, m.running_var)])
        off = 0
        for m in bns:
            c = m.num_features
            m.running_mean.data = run[off:off + c]
            m.running_var.data = run[off + c:off + 2 * c]
            off += 2 * c
        nbt = torch.stack([m.num_batches_tracked.detach() for m in bns]).to(torch.int64)
        for i, m in enumerate(bns):
            m.num_batches_tracked.data = nbt[i]
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_flat_run", run)
        object.__setattr__(self, "_flat_nbt", nbt)
        object.__setattr__(self, "_flat_grad", None)
        object.__setattr__(self, "_params", params)
        object.__setattr__(self, "_bns", bns)
        object.__setattr__(self, "_grads_attached", False)
        object.__setattr__(self, "_grads_stale", True)
        object.__setattr__(self, "_anchor", torch.zeros(1, device=dev, requires_grad=True))
        self.mark_params_changed()
        for p in params:
            p.grad = None

    def _storage_ok(self, full: bool = False) -> bool:
        """Are the parameters still views of the flat buffer?  Per-step check is O(1) (first / last parameter, last running
        statistic: `.to()`, `.float()`, `load_state_dict(assign=True)` move all of them); `full=True` walks every tensor."""
        flat = self._flat
        if flat is None:
            return False
        ps = self._params
        base = flat.data_ptr()
        last = self._bns[-1]
        if (ps[0].data_ptr() != base or ps[-1].data_ptr() != base + 4 * (flat.numel() - ps[-1].numel())
                or last.running_var.data_ptr() != self._flat_run.data_ptr() + 4 * (self._flat_run.numel() - last.num_features)):
            return False
        if full:
            off = 0
            for p in ps:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    return False
                off += p.numel()
        return True

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        object.__setattr__(self, "_flat", None)     # .to()/.cuda()/.float() re-allocate every tensor: re-flatten lazily
        return out

    # ---- plans --------------------------------------------------------------------------------------------------------
    def _plan_for(self, x: torch.Tensor):
        key = (tuple(x.shape), x.device.index)
        ent = self._plans.get(key)
        if ent is not None:
            self._plans.move_to_end(key)
            return ent
        L = _lib.lib()
        c = self.cfg
        if self.norm0.momentum is None:
            raise NotImplementedError("BatchNorm3d(momentum=None) (cumulative moving average) is not available in the fused backbone; "
                                      "the reference builds its norms with the default momentum 0.1 (models/densenet.py:198)")
        bc = list(c["block_config"]) + [0] * (8 - len(c["block_config"]))
        ccfg = _lib.DenseNetConfig(c["in_channels"], c["init_features"], c["growth_rate"], c["bn_size"], len(c["block_config"]),
                                   (ctypes.c_int32 * 8)(*bc), self.norm0.eps, self.norm0.momentum, c["dropout_prob"])
        n, _, d, h, w = x.shape
        plan = L.mmnn_densenet_plan_create(ctypes.byref(ccfg), n, d, h, w)
        if not plan:
            raise ValueError("mmnn_densenet_plan_create: " + _lib.last_error())
        if L.mmnn_densenet_param_count(plan) != self._flat.numel() or L.mmnn_densenet_runstat_count(plan) != self._flat_run.numel():
            L.mmnn_densenet_plan_destroy(plan)
            raise RuntimeError("parameter layout mismatch between the Python module tree and the native plan")
        shp = [ctypes.c_int32() for _ in range(4)]
        _lib.check(L.mmnn_densenet_out_shape(plan, *[ctypes.byref(v) for v in shp]), "out_shape")
        ws = torch.empty(L.mmnn_densenet_workspace_bytes(plan), dtype=torch.uint8, device=x.device)
        if os.environ.get("MMNN_POISON_WS") == "1":   # debugging aid: NaN-fill so reads of unwritten workspace words surface
            ws.fill_(255)
        # nn.BatchNorm3d.num_batches_tracked: the training forward's running-statistics kernel adds 1 to each (no torch op per step)
        _lib.check(L.mmnn_densenet_set_batch_counters(plan, self._flat_nbt.data_ptr(), self._flat_nbt.numel()), "set_batch_counters")
        ent = {"plan": plan, "ws": ws, "out_shape": (n,) + tuple(v.value for v in shp), "nbt_ptr": self._flat_nbt.data_ptr()}
        self._plans[key] = ent
        while len(self._plans) > self._MAX_PLANS:
            _, old = self._plans.popitem(last=False)
            L.mmnn_densenet_plan_destroy(old["plan"])
        return ent

    def __del__(self):
        try:
            L = _lib.lib()
            for ent in self._plans.values():
                L.mmnn_densenet_plan_destroy(ent["plan"])
        except Exception:
            pass

    # ---- execution ----------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("mmnn_sts_amd: the DenseNet backbone runs on the MI355X only (no CPU path); move model and input to cuda")
        if x.dim() != 5 or x.shape[1] != self.cfg["in_channels"]:
            raise ValueError(f"expected (N, {self.cfg['in_channels']}, D, H, W) input, got {tuple(x.shape)}")
        input_needs_grad = x.requires_grad
        x = x.detach()
        x = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        if not self._storage_ok():
            self._flatten()
        if self._flat.device != x.device:
            raise RuntimeError(f"model on {self._flat.device}, input on {x.device}")
        if self.training and torch.is_grad_enabled():
            return _BackboneFn.apply(x, self._anchor, self)
        out = self._run_forward(x, self.training)[0]
        if torch.is_grad_enabled() and not self.training and (input_needs_grad or any(p.requires_grad for p in self._params)):
            # eval-mode forward keeps no activations: a later backward() must fail loudly instead of silently producing no
            # backbone gradients (Grad-CAM has its own closed-form path: utils.MultiModalGradCAM).  A fully frozen encoder
            # (every backbone parameter requires_grad=False, input without grad) has no gradient to produce: its output is a
            # plain tensor, so heads / MLP / fusion layers can be trained on top of it as with the reference.
            return _NoBackward.apply(out, self._anchor)
        return out

    def _run_forward(self, x, training):
        ent = self._plan_for(x)
        out = torch.empty(ent["out_shape"], dtype=torch.float32, device=x.device)
        seed = ops.next_seed()
        if ent["nbt_ptr"] != self._flat_nbt.data_ptr():          # re-flattened since the plan was made
            _lib.check(_lib.lib().mmnn_densenet_set_batch_counters(ent["plan"], self._flat_nbt.data_ptr(), self._flat_nbt.numel()), "set_batch_counters")
            ent["nbt_ptr"] = self._flat_nbt.data_ptr()
        version = self._params_version() if self.repack_policy == "versioned" else 0
        if version != ent.get("version", 0):
            _lib.check(_lib.lib().mmnn_densenet_set_option(ent["plan"], b"params_version", version), "set_option")
            ent["version"] = version
        _lib.check(_lib.lib().mmnn_densenet_forward(ent["plan"], self._flat.data_ptr(), self._flat_run.data_ptr(), x.data_ptr(),
                                                    ent["ws"].data_ptr(), out.data_ptr(), int(training), seed,
                                                    torch.cuda.current_stream().cuda_stream), "mmnn_densenet_forward")
        object.__setattr__(self, "_fwd_token", self._fwd_token + 1)
        return out, ent, seed, self._fwd_token

    def _run_backward(self, x, ent, seed, token, grad_out):
        if token != self._fwd_token:
            raise RuntimeError("mmnn_sts_amd: backward through a DenseNet backbone forward whose workspace has been overwritten by a "
                               "later forward of the same module (one live forward per module is supported)")
        params = self._params
        gflat = self._flat_grad
        if gflat is None:
            gflat = torch.empty_like(self._flat)
            object.__setattr__(self, "_flat_grad", gflat)
            object.__setattr__(self, "_grads_attached", False)
        g0 = params[0].grad
        attached = self._grads_attached and g0 is not None and g0.data_ptr() == gflat.data_ptr()
        # overwrite when nothing is accumulated yet: gradients dropped by zero_grad(set_to_none=True) of any optimizer, or marked
        # stale by FusedSGD.zero_grad() (which keeps the views attached: re-attaching 364 tensors per step costs ~1.5 ms of host time)
        fresh = (not attached) or self._grads_stale
        hook = self._grad_ready_hook
        st = torch.cuda.current_stream().cuda_stream
        if hook is None:
            _lib.check(_lib.lib().mmnn_densenet_backward(ent["plan"], self._flat.data_ptr(), x.data_ptr(), ent["ws"].data_ptr(),
                                                         grad_out.data_ptr(), gflat.data_ptr(), 0 if fresh else 1, seed, st),
                       "mmnn_densenet_backward")
        else:
            for blk, begin, end in self.block_param_ranges(ent):
                _lib.check(_lib.lib().mmnn_densenet_backward_range(ent["plan"], self._flat.data_ptr(), x.data_ptr(), ent["ws"].data_ptr(),
                                                                   grad_out.data_ptr(), gflat.data_ptr(), 0 if fresh else 1, seed, blk, blk, st),
                           "mmnn_densenet_backward_range")
                hook(self, begin, end)
        if not attached:   # (re)attach .grad views; afterwards gradients accumulate inside the kernel
            off = 0
            for p in params:
                n = p.numel()
                p.grad = gflat[off:off + n].view(p.shape)
                off += n
            object.__setattr__(self, "_grads_attached", True)
        object.__setattr__(self, "_grads_stale", False)

    def mark_grads_stale(self) -> None:
        """The next backward overwrites the flat gradient buffer instead of accumulating (cheap zero_grad)."""
        object.__setattr__(self, "_grads_stale", True)

    # flat views for the fused optimizer / gradient all-reduce
    @property
    def flat_parameters(self) -> torch.Tensor:
        if not self._storage_ok():
            self._flatten()
        return self._flat

    @property
    def flat_grad(self):
        return self._flat_grad


class _BackboneFn(torch.autograd.Function):
    """Autograd node of the whole backbone.  Parameter gradients are written by the HIP backward straight into the
    module's flat gradient buffer (the `.grad` of every parameter is a view of it), so nothing is returned for them."""

    @staticmethod
    def forward(ctx, x, anchor, module):
        out, ent, seed, token = module._run_forward(x, True)
        ctx.module, ctx.ent, ctx.seed, ctx.token = module, ent, seed, token
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        g = grad_out if (grad_out.dtype == torch.float32 and grad_out.is_contiguous()) else grad_out.float().contiguous()
        ctx.module._run_backward(x, ctx.ent, ctx.seed, ctx.token, g)
        return None, None, None


class _NoBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, anchor):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad_out):
        raise RuntimeError("mmnn_sts_amd: backward through an eval-mode DenseNet backbone forward is not supported (no activations "
                           "are kept in eval mode); call model.train() for training, wrap inference in torch.no_grad(), or use "
                           "utils.MultiModalGradCAM for attention maps")


class _Features(nn.Sequential):
    """relu -> AdaptiveAvgPool3d(1) -> flatten -> feature_layer -> dropout (models/densenet.py:234-247), one fused op."""

    def __init__(self, in_channels: int, feature_channels: int, dropout_prob: float):
        super().__init__(OrderedDict([
            ("relu", nn.ReLU(inplace=True)), ("pool", nn.AdaptiveAvgPool3d(1)), ("flatten", nn.Flatten(1)),
            ("feature_layer", nn.Linear(in_channels, feature_channels)), ("dropout", nn.Dropout(dropout_prob)),
        ]))

    def forward(self, h: torch.Tensor) -> torch.Tensor:
        return ops.GapLinear.apply(h, self.feature_layer.weight, self.feature_layer.bias, float(self.dropout.p), self.training)


class _ClassLayers(nn.Sequential):
    def __init__(self, feature_channels: int, out_channels: int):
        super().__init__(OrderedDict([("out", nn.Linear(feature_channels, out_channels))]))

    def forward(self, f: torch.Tensor) -> torch.Tensor:
        return ops.SmallLinear.apply(f, self.out.weight, self.out.bias)


class DenseNet(nn.Module):
    """Drop-in for `models.densenet.DenseNet` (models/densenet.py:151-271); 3-D, batch norm, ReLU only."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, feature_channels: int, init_features: int = 64,
                 growth_rate: int = 32, block_config: Sequence[int] = (6, 12, 24, 16), bn_size: int = 4,
                 act: Union[str, tuple] = ("relu", {"inplace": True}), norm: Union[str, tuple] = "batch",
                 dropout_prob: float = 0.0) -> None:
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("mmnn_sts_amd implements the 3-D DenseNet of the fusion path only (spatial_dims=3)")
        if (act if isinstance(act, str) else act[0]).lower() != "relu" or (norm if isinstance(norm, str) else norm[0]).lower() != "batch":
            raise NotImplementedError("mmnn_sts_amd kernels fuse ReLU + batch norm; other act/norm choices are not available")
        # limits of the native kernels (csrc/densenet.hip: plan_build), reported when the module is built rather than at its first forward
        if not (1 <= int(in_channels) <= 4):
            raise ValueError(f"mmnn_sts_amd: in_channels must be 1..4 (the stem kernel's input tile), got {in_channels}")
        if not (1 <= int(init_features) <= 64) or not (1 <= int(growth_rate) <= 32):
            raise ValueError(f"mmnn_sts_amd: init_features <= 64 and growth_rate <= 32 (one MFMA row tile each), got {init_features} / {growth_rate}")
        if not (1 <= len(block_config) <= 8) or any(int(n) < 1 for n in block_config) or int(bn_size) < 1 or not (0.0 <= float(dropout_prob) < 1.0):
            raise ValueError(f"mmnn_sts_amd: bad block_config / bn_size / dropout_prob: {block_config} / {bn_size} / {dropout_prob}")
        self.backbone = _Backbone(in_channels, init_features, growth_rate, block_config, bn_size, dropout_prob)
        self.features = _Features(self.backbone.out_channels, feature_channels, dropout_prob)
        self.class_layers = _ClassLayers(feature_channels, out_channels)
        for m in self.modules():      # models/densenet.py:258-265
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.class_layers(self.features(self.backbone(x)))


class DenseNet121(DenseNet):
    """models/densenet.py:312-331 (no pretrained 3-D weights exist upstream either)."""

    def __init__(self, init_features: int = 64, growth_rate: int = 32, block_config: Sequence[int] = (6, 12, 24, 16),
                 pretrained: bool = False, progress: bool = True, **kwargs) -> None:
        super().__init__(init_features=init_features, growth_rate=growth_rate, block_config=block_config, **kwargs)
        if pretrained:
            raise NotImplementedError("Parameter `spatial_dims` is > 2 ; PyTorch Hub provides no pretrained 3-D DenseNet")


class TinyDensenet(DenseNet):
    """models/densenet.py:333-356: block_config (6, 12, 4)."""

    def __init__(self, init_features: int = 64, growth_rate: int = 32, block_config: Sequence[int] = (6, 12, 4),
                 pretrained: bool = False, progress: bool = True, **kwargs) -> None:
        super().__init__(init_features=init_features, growth_rate=growth_rate, block_config=block_config, **kwargs)
        if pretrained:
            raise NotImplementedError("Parameter `spatial_dims` is > 2 ; PyTorch Hub provides no pretrained 3-D DenseNet")
