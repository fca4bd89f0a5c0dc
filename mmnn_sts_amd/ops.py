"""torch.autograd.Function wrappers over the C-ABI (include/mmnn_sts.h).  PyTorch tensors are storage only: every
forward/backward below is one or two launches of hand-written HIP kernels; there is no eager / CPU fallback."""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mmnn_sts_amd: tensors must live on the MI355X (cuda) device; there is no CPU path")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_seed_counter = [0]


def next_seed() -> int:
    """Per-call dropout stream id, derived from torch's seed so `torch.manual_seed` makes runs repeatable, and from the
    data-parallel rank so that ranks seeded alike still draw different masks."""
    _seed_counter[0] += 1
    rank = int(os.environ.get("RANK", "0") or 0)
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0] * 0xD1B54A32D192ED03
            + rank * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------------------------------------------------
# DenseNet.features  (models/densenet.py:234-247)
# ----------------------------------------------------------------------------------------------------------------------
class GapLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, weight, bias, p: float, training: bool):
        _need_cuda(h, weight, bias)
        h, weight, bias = _f32c(h), _f32c(weight), _f32c(bias)
        n, c = h.shape[0], h.shape[1]
        v = h[0, 0].numel()
        f = weight.shape[0]
        pooled = torch.empty((n, c), device=h.device, dtype=torch.float32)
        out = torch.empty((n, f), device=h.device, dtype=torch.float32)
        seed = next_seed()
        _lib.check(_lib.lib().mmnn_gap_linear_forward(n, c, v, f, h.data_ptr(), weight.data_ptr(), bias.data_ptr(), pooled.data_ptr(),
                                                      out.data_ptr(), float(p), seed, int(training), _stream()), "gap_linear_forward")
        ctx.save_for_backward(h, weight, pooled)
        ctx.meta = (n, c, v, f, float(p), seed, int(training))
        return out

    @staticmethod
    def backward(ctx, dout):
        h, weight, pooled = ctx.saved_tensors
        n, c, v, f, p, seed, training = ctx.meta
        dout = _f32c(dout)
        dw = torch.empty_like(weight)
        db = torch.empty((f,), device=h.device, dtype=torch.float32)
        dh = torch.empty_like(h)
        _lib.check(_lib.lib().mmnn_gap_linear_backward(n, c, v, f, h.data_ptr(), weight.data_ptr(), pooled.data_ptr(), dout.data_ptr(),
                                                       dw.data_ptr(), db.data_ptr(), dh.data_ptr(), p, seed, training, 0, _stream()),
                   "gap_linear_backward")
        return dh, dw, db, None, None


# ----------------------------------------------------------------------------------------------------------------------
# [Linear -> BatchNorm1d -> ReLU/Dropout1d] stacks  (models/mlp.py:19-51)
# ----------------------------------------------------------------------------------------------------------------------
def _mlp_desc(n, dims_in, dims_out, relu_first, p, eps, momentum, seed, training, first_id):
    d = _lib.MlpDesc()
    d.n, d.num_layers = n, len(dims_in)
    for i, (a, b, r) in enumerate(zip(dims_in, dims_out, relu_first)):
        d.in_dim[i], d.out_dim[i], d.relu_first[i] = a, b, int(r)
    d.dropout_prob, d.eps, d.momentum, d.seed, d.training, d.first_layer_id = float(p), float(eps), float(momentum), seed, int(training), first_id
    return d


class MlpStack(torch.autograd.Function):
    """inputs: x, then per layer (weight, bias, gamma, beta); running statistics are updated in place by the kernel."""

    @staticmethod
    def forward(ctx, x, cfg, running, *params):
        relu_first, p, eps, momentum, training, first_id = cfg[:6]
        _need_cuda(x, *params)
        x = _f32c(x)
        nl = len(params) // 4
        ws = [_f32c(t) for t in params]
        dims_in = [ws[4 * i].shape[1] for i in range(nl)]
        dims_out = [ws[4 * i].shape[0] for i in range(nl)]
        n = x.shape[0]
        if x.dim() != 2 or x.shape[1] != dims_in[0]:
            raise ValueError(f"MLP stack expects (N, {dims_in[0]}) input, got {tuple(x.shape)}")
        if training and n == 1:     # torch.nn.BatchNorm1d refuses this too (models/mlp.py:22-48 in training mode)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        seed = next_seed()
        desc = _mlp_desc(n, dims_in, dims_out, relu_first, p, eps, momentum, seed, training, first_id)
        pp = _lib.MlpParams()
        for i in range(nl):
            pp.weight[i], pp.bias[i], pp.gamma[i], pp.beta[i] = (ws[4 * i + j].data_ptr() for j in range(4))
            pp.running_mean[i], pp.running_var[i] = running[2 * i].data_ptr(), running[2 * i + 1].data_ptr()
            if len(cfg) > 6 and cfg[6] is not None:           # nn.BatchNorm1d.num_batches_tracked, +1 per training forward (in the kernel)
                pp.num_batches_tracked[i] = cfg[6][i].data_ptr()
        L = _lib.lib()
        saved = torch.empty((max(1, L.mmnn_mlp_saved_floats(ctypes.byref(desc))),), device=x.device, dtype=torch.float32)
        out = torch.empty((n, dims_out[-1]), device=x.device, dtype=torch.float32)
        _lib.check(L.mmnn_mlp_forward(ctypes.byref(desc), ctypes.byref(pp), x.data_ptr(), out.data_ptr(), saved.data_ptr(), _stream()),
                   "mlp_forward")
        ctx.save_for_backward(x, saved, *ws)
        ctx.desc_args = (n, dims_in, dims_out, relu_first, p, eps, momentum, seed, training, first_id)
        ctx.running = running
        return out

    @staticmethod
    def backward(ctx, dy):
        x, saved, *ws = ctx.saved_tensors
        n, dims_in, dims_out = ctx.desc_args[:3]
        nl = len(dims_in)
        desc = _mlp_desc(*ctx.desc_args)
        pp = _lib.MlpParams()
        grads = [torch.empty_like(w) for w in ws]
        for i in range(nl):
            pp.weight[i], pp.bias[i], pp.gamma[i], pp.beta[i] = (ws[4 * i + j].data_ptr() for j in range(4))
            pp.running_mean[i], pp.running_var[i] = ctx.running[2 * i].data_ptr(), ctx.running[2 * i + 1].data_ptr()
            pp.grad_weight[i], pp.grad_bias[i], pp.grad_gamma[i], pp.grad_beta[i] = (grads[4 * i + j].data_ptr() for j in range(4))
        dy = _f32c(dy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        scratch = torch.empty((2 * n * max(dims_in + dims_out),), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_mlp_backward(ctypes.byref(desc), ctypes.byref(pp), x.data_ptr(), saved.data_ptr(), dy.data_ptr(),
                                                dx.data_ptr() if dx is not None else None, scratch.data_ptr(), 0, _stream()), "mlp_backward")
        return (dx, None, None, *grads)


# ----------------------------------------------------------------------------------------------------------------------
# small dense layer
# ----------------------------------------------------------------------------------------------------------------------
class SmallLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_cuda(x, weight, bias)
        x, weight = _f32c(x), _f32c(weight)
        bias = _f32c(bias) if bias is not None else None
        n, d, o = x.shape[0], x.shape[1], weight.shape[0]
        y = torch.empty((n, o), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_linear_forward(n, d, o, x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                  y.data_ptr(), _stream()), "linear_forward")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32c(dy)
        n, d, o = x.shape[0], x.shape[1], weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight)
        db = torch.empty((o,), device=x.device, dtype=torch.float32) if ctx.has_bias else None
        _lib.check(_lib.lib().mmnn_linear_backward(n, d, o, x.data_ptr(), weight.data_ptr(), dy.data_ptr(),
                                                   dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                                   db.data_ptr() if db is not None else None, 0, _stream()), "linear_backward")
        return dx, dw, db


# ----------------------------------------------------------------------------------------------------------------------
# fusion heads  (models/multimodal.py:62-77)
# ----------------------------------------------------------------------------------------------------------------------
class FusionHeads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fi, fc, wf, bf, wi, bi, wc, bc, blend: bool):
        _need_cuda(fi, fc, wf, bf, wi, bi, wc, bc)
        fi, fc, wf, bf, wi, bi, wc, bc = (_f32c(t) for t in (fi, fc, wf, bf, wi, bi, wc, bc))
        n, f = fi.shape
        c = wf.shape[0]
        out = torch.empty((3, n, c) if blend else (n, c), device=fi.device, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_fusion_heads_forward(n, f, c, int(blend), fi.data_ptr(), fc.data_ptr(), wf.data_ptr(), bf.data_ptr(),
                                                        wi.data_ptr(), bi.data_ptr(), wc.data_ptr(), bc.data_ptr(), out.data_ptr(),
                                                        _stream()), "fusion_heads_forward")
        ctx.save_for_backward(fi, fc, wf, wi, wc)
        ctx.meta = (n, f, c, bool(blend))
        return out

    @staticmethod
    def backward(ctx, dout):
        fi, fc, wf, wi, wc = ctx.saved_tensors
        n, f, c, blend = ctx.meta
        dout = _f32c(dout)
        dev = fi.device
        dfi, dfc = torch.empty_like(fi), torch.empty_like(fc)
        dwf = torch.empty_like(wf)
        dbf = torch.empty((c,), device=dev, dtype=torch.float32)
        # the per-modality heads take part in the graph only with blend (models/multimodal.py:69); the kernel then writes them in full
        dwi, dwc = (torch.empty_like(wi), torch.empty_like(wc)) if blend else (None, None)
        dbi, dbc = (torch.empty((c,), device=dev, dtype=torch.float32), torch.empty((c,), device=dev, dtype=torch.float32)) if blend else (None, None)
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(_lib.lib().mmnn_fusion_heads_backward(n, f, c, int(blend), fi.data_ptr(), fc.data_ptr(), wf.data_ptr(), wi.data_ptr(),
                                                         wc.data_ptr(), dout.data_ptr(), dfi.data_ptr(), dfc.data_ptr(), dwf.data_ptr(),
                                                         dbf.data_ptr(), ptr(dwi), ptr(dbi), ptr(dwc), ptr(dbc), 0,
                                                         _stream()), "fusion_heads_backward")
        return dfi, dfc, dwf, dbf, dwi, dbi, dwc, dbc, None


# ----------------------------------------------------------------------------------------------------------------------
# Cox partial likelihood, summed over targets, blended over heads
# ----------------------------------------------------------------------------------------------------------------------
class CoxBlend(torch.autograd.Function):
    """preds (H, N, C); sort_key / weight (N, C) in pycox's (durations, events) positions, any integer or floating dtype
    (the reference's datasets produce int64 and float32; both travel as fp64, which holds them exactly -- fractional
    durations are NOT truncated).  Returns (loss, head_losses) with loss = sum_h head_weights[h] * head_losses[h]."""

    @staticmethod
    def forward(ctx, preds, sort_key, weight, head_weights):
        _need_cuda(preds, sort_key, weight, head_weights)
        preds = _f32c(preds)
        sort_key, kdt = _cox_operand(sort_key)
        weight, wdt = _cox_operand(weight)
        h, n, c = preds.shape
        dev = preds.device
        hw = _f32c(head_weights) if head_weights is not None else None
        out = torch.empty((1 + h,), device=dev, dtype=torch.float32)
        grad = torch.empty_like(preds)
        scratch = torch.empty((4 * n,), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_cox_blend_loss_typed(h, n, c, preds.data_ptr(), sort_key.data_ptr(), kdt, weight.data_ptr(), wdt,
                                                        hw.data_ptr() if hw is not None else None, out.data_ptr(), out[1:].data_ptr(),
                                                        grad.data_ptr(), scratch.data_ptr(), _stream()), "cox_blend_loss")
        ctx.save_for_backward(grad, hw)
        ctx.set_materialize_grads(False)     # unused outputs arrive as None instead of zero tensors (no device sync needed)
        return out[0], out[1:]

    @staticmethod
    def backward(ctx, dloss, dheads):
        grad, hw = ctx.saved_tensors
        if dloss is None and dheads is None:
            return None, None, None, None
        h, n, c = grad.shape
        g = torch.empty_like(grad)
        dl = _f32c(dloss) if dloss is not None else None
        dh = _f32c(dheads) if dheads is not None else None      # d head_losses[h] / d preds = grad[h] / head_weights[h]
        _lib.check(_lib.lib().mmnn_cox_blend_backward(h, n, c, grad.data_ptr(), hw.data_ptr() if hw is not None else None,
                                                      dl.data_ptr() if dl is not None else None, dh.data_ptr() if dh is not None else None,
                                                      g.data_ptr(), _stream()), "cox_blend_backward")
        return g, None, None, None


_COX_DTYPES = {torch.float64: _lib.DT_F64, torch.float32: _lib.DT_F32, torch.int64: _lib.DT_I64, torch.int32: _lib.DT_I32, torch.uint8: _lib.DT_U8,
               torch.bool: _lib.DT_U8}


def _cox_operand(t: torch.Tensor):
    """(contiguous tensor, MMNN_DT_* code): the kernel reads int64 / int32 / uint8 / bool / float32 / float64 directly (all exact in its
    fp64 arithmetic); anything else (float16, int16, ...) is converted to float64 first."""
    if t.dtype not in _COX_DTYPES:
        t = t.to(torch.float64)
    return (t if t.is_contiguous() else t.contiguous()), _COX_DTYPES[t.dtype]


# ----------------------------------------------------------------------------------------------------------------------
# element-wise BCE on logits with positive-class weights  (classification trainer, main.py:147-153)
# ----------------------------------------------------------------------------------------------------------------------
class BceLogits(torch.autograd.Function):
    """loss[...] = pw_c * y * softplus(-x) + (1 - y) * softplus(x), class axis last; no reduction."""

    @staticmethod
    def forward(ctx, logits, targets, pos_weight):
        _need_cuda(logits, targets, pos_weight)
        x = _f32c(logits)
        y = _f32c(targets.expand_as(logits) if targets.shape != logits.shape else targets)
        c = x.shape[-1]
        pw = _f32c(pos_weight) if pos_weight is not None else None
        if pw is not None and pw.numel() != c:
            raise ValueError(f"pos_weight has {pw.numel()} entries for {c} classes")
        loss = torch.empty_like(x)
        dldx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.lib().mmnn_bce_logits(x.numel(), c, x.data_ptr(), y.data_ptr(), pw.data_ptr() if pw is not None else None,
                                              loss.data_ptr(), dldx.data_ptr() if dldx is not None else None, _stream()), "bce_logits")
        ctx.save_for_backward(dldx)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dldx,) = ctx.saved_tensors
        return (dloss * dldx if dldx is not None else None), None, None


# ----------------------------------------------------------------------------------------------------------------------
# 3-D ResNet-18 variant  (models/resnet.py:5-227): direct convolution, BN [+ residual] [+ ReLU] [+ dropout], pooled sigmoid head
# ----------------------------------------------------------------------------------------------------------------------
def _triple(v):
    return tuple(int(t) for t in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def _conv_desc(x_shape, c_out, kernel, stride, padding):
    n, c, d, h, w = x_shape
    return _lib.Conv3dDesc(n, c, d, h, w, c_out, (ctypes.c_int32 * 3)(*kernel), (ctypes.c_int32 * 3)(*stride), (ctypes.c_int32 * 3)(*padding))


class Conv3dDirect(torch.autograd.Function):
    """y = conv3d(x, w, stride, padding), no bias, any kernel extent (Conv3DSimple, BasicStem conv, downsample conv)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        _need_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        if x.dim() != 5 or weight.dim() != 5 or x.shape[1] != weight.shape[1]:
            raise ValueError(f"conv3d: input {tuple(x.shape)} does not match weight {tuple(weight.shape)}")
        stride, padding = _triple(stride), _triple(padding)
        desc = _conv_desc(x.shape, weight.shape[0], tuple(weight.shape[2:]), stride, padding)
        L = _lib.lib()
        o = [ctypes.c_int32() for _ in range(3)]
        _lib.check(L.mmnn_conv3d_out_shape(ctypes.byref(desc), *[ctypes.byref(v) for v in o]), "conv3d_out_shape")
        y = torch.empty((x.shape[0], weight.shape[0], o[0].value, o[1].value, o[2].value), device=x.device, dtype=torch.float32)
        _lib.check(L.mmnn_conv3d_forward(ctypes.byref(desc), x.data_ptr(), weight.data_ptr(), y.data_ptr(), _stream()), "conv3d_forward")
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding = ctx.geom
        dy = _f32c(dy)
        desc = _conv_desc(x.shape, weight.shape[0], tuple(weight.shape[2:]), stride, padding)
        L = _lib.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.mmnn_conv3d_backward_data(ctypes.byref(desc), dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), _stream()), "conv3d_backward_data")
        dw = torch.empty_like(weight)
        ws = torch.empty((L.mmnn_conv3d_wgrad_workspace_bytes(ctypes.byref(desc)),), dtype=torch.uint8, device=x.device)
        _lib.check(L.mmnn_conv3d_backward_weight(ctypes.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 0, _stream()),
                   "conv3d_backward_weight")
        return dx, dw, None, None


class BatchNormAct3d(torch.autograd.Function):
    """out = dropout(relu?(BN(x) [+ residual])); running statistics are updated in place in training mode."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, momentum, eps, training, relu, drop_p):
        _need_cuda(x, gamma, beta, running_mean, running_var, residual)
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        res = _f32c(residual) if residual is not None else None
        if res is not None and res.shape != x.shape:
            raise ValueError(f"residual {tuple(res.shape)} does not match {tuple(x.shape)}")
        n, c = x.shape[0], x.shape[1]
        v = x[0, 0].numel()
        out = torch.empty_like(x)
        save = torch.empty((2, c), device=x.device, dtype=torch.float32)
        stat = torch.empty((2, c), device=x.device, dtype=torch.float64)
        seed = next_seed()
        _lib.check(_lib.lib().mmnn_bn3d_forward(n, c, v, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                                running_var.data_ptr(), float(momentum), float(eps), int(training), int(relu),
                                                res.data_ptr() if res is not None else None, float(drop_p), seed, out.data_ptr(),
                                                save.data_ptr(), stat.data_ptr(), _stream()), "bn3d_forward")
        ctx.save_for_backward(x, out, gamma, save)
        ctx.meta = (int(training), int(relu), float(drop_p), seed, res is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gamma, save = ctx.saved_tensors
        training, relu, drop_p, seed, has_res = ctx.meta
        dout = _f32c(dout)
        n, c = x.shape[0], x.shape[1]
        v = x[0, 0].numel()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and ctx.needs_input_grad[5]) else None
        dg = torch.empty((c,), device=x.device, dtype=torch.float32)
        db = torch.empty((c,), device=x.device, dtype=torch.float32)
        stat = torch.empty((2, c), device=x.device, dtype=torch.float64)
        _lib.check(_lib.lib().mmnn_bn3d_backward(n, c, v, x.data_ptr(), out.data_ptr(), dout.data_ptr(), gamma.data_ptr(), save.data_ptr(),
                                                 training, relu, drop_p, seed, dx.data_ptr(), dres.data_ptr() if dres is not None else None,
                                                 dg.data_ptr(), db.data_ptr(), stat.data_ptr(), _stream()), "bn3d_backward")
        return dx, dg, db, None, None, dres, None, None, None, None, None


class GapFcSigmoid(torch.autograd.Function):
    """sigmoid(Linear(mean over voxels))  (models/resnet.py:152-167)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_cuda(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        n, c = x.shape[0], x.shape[1]
        v = x[0, 0].numel()
        o = weight.shape[0]
        pooled = torch.empty((n, c), device=x.device, dtype=torch.float32)
        y = torch.empty((n, o), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_gap_fc_sigmoid_forward(n, c, v, o, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), pooled.data_ptr(),
                                                          y.data_ptr(), _stream()), "gap_fc_sigmoid_forward")
        ctx.save_for_backward(weight, pooled, y)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, pooled, y = ctx.saved_tensors
        dy = _f32c(dy)
        n, c = ctx.shape[0], ctx.shape[1]
        v = 1
        for t in ctx.shape[2:]:
            v *= t
        dx = torch.empty(ctx.shape, device=weight.device, dtype=torch.float32)
        dw = torch.empty_like(weight)
        db = torch.empty((weight.shape[0],), device=weight.device, dtype=torch.float32)
        _lib.check(_lib.lib().mmnn_gap_fc_sigmoid_backward(n, c, v, weight.shape[0], weight.data_ptr(), pooled.data_ptr(), y.data_ptr(),
                                                           dy.data_ptr(), dw.data_ptr(), db.data_ptr(), dx.data_ptr(), _stream()),
                   "gap_fc_sigmoid_backward")
        return dx, dw, db
