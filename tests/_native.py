"""Direct ctypes driver of the backbone C-ABI for white-box GPU tests (no nn.Module layer involved)."""
import ctypes

import numpy as np
import torch

from mmnn_sts_amd import _lib
from oracle import restatement as R


def make_cfg(cfg: R.DenseNetCfg, dropout=0.0):
    bc = list(cfg.block_config) + [0] * (8 - len(cfg.block_config))
    return _lib.DenseNetConfig(cfg.in_channels, cfg.init_features, cfg.growth_rate, cfg.bn_size, len(cfg.block_config),
                               (ctypes.c_int32 * 8)(*bc), 1e-5, 0.1, dropout)


def backbone_param_keys(schema):
    return [k for k in schema if k.startswith("backbone.") and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def backbone_run_keys(schema):
    return [k for k in schema if k.startswith("backbone.") and k.endswith(("running_mean", "running_var"))]


class NativeBackbone:
    def __init__(self, cfg: R.DenseNetCfg, n, d, h, w, dropout=0.0, device="cuda"):
        self.L = _lib.lib()
        self.cfg = cfg
        self.ccfg = make_cfg(cfg, dropout)
        self.plan = self.L.mmnn_densenet_plan_create(ctypes.byref(self.ccfg), n, d, h, w)
        if not self.plan:
            raise ValueError(_lib.last_error())
        self.n_params = self.L.mmnn_densenet_param_count(self.plan)
        self.n_run = self.L.mmnn_densenet_runstat_count(self.plan)
        self.ws_bytes = self.L.mmnn_densenet_workspace_bytes(self.plan)
        c = [ctypes.c_int32() for _ in range(4)]
        _lib.check(self.L.mmnn_densenet_out_shape(self.plan, *[ctypes.byref(v) for v in c]), "out_shape")
        self.out_shape = (n,) + tuple(v.value for v in c)
        self.device = device
        self.in_dhw = (d, h, w)
        # adversarial fill: every fp32 / fp64 word of the workspace starts as NaN, so a kernel that reads a word nobody wrote shows up
        self.ws = torch.full((self.ws_bytes,), 255, dtype=torch.uint8, device=device)
        self.schema = R.densenet_schema(cfg)

    def __del__(self):
        try:
            self.L.mmnn_densenet_plan_destroy(self.plan)
        except Exception:
            pass

    def flatten(self, sd):
        pk = backbone_param_keys(self.schema)
        rk = backbone_run_keys(self.schema)
        flat = torch.cat([sd[k].detach().reshape(-1).float() for k in pk]).to(self.device)
        run = torch.cat([sd[k].detach().reshape(-1).float() for k in rk]).to(self.device)
        assert flat.numel() == self.n_params and run.numel() == self.n_run
        return flat.contiguous(), run.contiguous()

    def unflatten(self, flat, keys=None):
        out, o = {}, 0
        for k in (keys or backbone_param_keys(self.schema)):
            n = int(np.prod(self.schema[k]))
            out[k] = flat[o:o + n].view(self.schema[k])
            o += n
        return out

    def forward(self, flat, run, x, training=True, seed=0):
        out = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(self.L.mmnn_densenet_forward(self.plan, flat.data_ptr(), run.data_ptr(), x.data_ptr(), self.ws.data_ptr(),
                                                out.data_ptr(), int(training), seed, st), "densenet_forward")
        return out

    def backward(self, flat, x, grad_out, accumulate=False, seed=0, grad=None):
        if grad is None:
            grad = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(self.L.mmnn_densenet_backward(self.plan, flat.data_ptr(), x.data_ptr(), self.ws.data_ptr(), grad_out.data_ptr(),
                                                 grad.data_ptr(), int(accumulate), seed, st), "densenet_backward")
        return grad

    def region(self, name, shape, i=0, j=0, dtype=torch.float32):
        off = self.L.mmnn_densenet_ws_offset(self.plan, name.encode(), i, j)
        assert off >= 0, name
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.ws[off:off + n].view(dtype).view(shape)

    def relu_masks(self, flat):
        """All ReLU branch decisions of the last training forward, keyed like oracle.restatement.densenet_backbone."""
        n = self.out_shape[0]
        st = torch.cuda.current_stream().cuda_stream
        masks = {}

        def fetch(kind, b, l, shape):
            m = torch.empty(shape, dtype=torch.uint8, device=self.device)
            _lib.check(self.L.mmnn_densenet_relu_mask(self.plan, flat.data_ptr(), self.ws.data_ptr(), kind, b, l, m.data_ptr(), st), "relu_mask")
            return m.cpu()

        cfg = self.cfg
        d0 = [(s - 1) // 2 + 1 for s in self.in_dhw]
        masks["relu0"] = fetch(0, 0, 0, (n, cfg.init_features, *d0))
        dims = [(s - 1) // 2 + 1 for s in d0]
        c = cfg.init_features
        mid = cfg.bn_size * cfg.growth_rate
        for b, nl in enumerate(cfg.block_config):
            for l in range(nl):
                masks[f"b{b + 1}l{l + 1}r1"] = fetch(1, b, l, (n, c, *dims))
                masks[f"b{b + 1}l{l + 1}r2"] = fetch(2, b, l, (n, mid, *dims))
                c += cfg.growth_rate
            if b != len(cfg.block_config) - 1:
                masks[f"t{b + 1}"] = fetch(3, b, 0, (n, c, *dims))
                c //= 2
                dims = [s // 2 for s in dims]
        return masks
