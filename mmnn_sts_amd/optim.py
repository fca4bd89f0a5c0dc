"""Fused SGD (momentum / Nesterov / weight decay) for the fusion model -- the optimizer of main.py:410-413.

torch.optim.SGD works unchanged on the native modules (their parameters are ordinary nn.Parameters); this class does the
same update with TWO HIP launches per step: one over each backbone's flat parameter buffer (11.26 M of the 11.28 M parameters,
`mmnn_sgd_step`) and one over the list of the ~30 small tensors outside it (`mmnn_sgd_step_multi`: MLP, feature layer, heads; the
pointer table travels as a kernel argument).  It is a torch.optim.Optimizer, so lr schedulers (OneCycleLR, which also cycles
`momentum`) drive it through `param_groups` as usual.  Parameters whose `.grad` is None are skipped like torch.optim.SGD skips them
(no weight decay, no momentum buffer: the four never-trained tensors of SURVEY A6 stay untouched).
"""
import torch

from . import _lib


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, model: torch.nn.Module, lr=1e-3, momentum=0.0, nesterov=False, weight_decay=0.0):
        from .models.densenet import _Backbone
        self._backbones = [m for m in model.modules() if isinstance(m, _Backbone)]
        for bb in self._backbones:
            bb.flat_parameters  # force the flat layout before the parameter list is captured
        flat_ids = {id(p) for bb in self._backbones for p in bb.parameters()}
        self._rest = [p for p in model.parameters() if id(p) not in flat_ids]
        super().__init__(list(model.parameters()), dict(lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=weight_decay))
        self._bufs = {}
        # momentum buffers of the small tensors: ONE flat buffer, tensor i at _rest_off[i]; a tensor's buffer starts with its first step
        self._rest_off, total = [], 0
        for p in self._rest:
            self._rest_off.append(total)
            total += (p.numel() + 3) // 4 * 4
        self._rest_total = total
        self._rest_buf = None
        self._rest_started = set()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        lr, mom, wd, nes = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), int(bool(g["nesterov"]))
        st = torch.cuda.current_stream().cuda_stream
        L = _lib.lib()
        for bb in self._backbones:
            flat, grad = bb.flat_parameters, bb.flat_grad
            if grad is None or bb.conv0.weight.grad is None or bb._grads_stale:
                continue          # no gradient since the last zero_grad()
            first = id(bb) not in self._bufs or self._bufs[id(bb)].data_ptr() == 0 or self._bufs[id(bb)].numel() != flat.numel()
            if first:
                self._bufs[id(bb)] = torch.empty_like(flat)
            _lib.check(L.mmnn_sgd_step(flat.data_ptr(), grad.data_ptr(), self._bufs[id(bb)].data_ptr(), flat.numel(), lr, mom, wd,
                                       nes, int(first), st), "sgd_step")
            bb.mark_params_changed()          # written through the raw pointer: invisible to autograd's version counters
        live = [(i, p) for i, p in enumerate(self._rest) if p.grad is not None]
        if live:
            dev = live[0][1].device
            if self._rest_buf is None or self._rest_buf.device != dev:
                self._rest_buf = torch.zeros((max(1, self._rest_total),), device=dev, dtype=torch.float32)
                self._rest_started = set()
            for lo in range(0, len(live), _lib.MULTI_MAX):
                chunk = live[lo:lo + _lib.MULTI_MAX]
                refs = (_lib.TensorRef * len(chunk))()
                for r, (i, p) in zip(refs, chunk):
                    gr = p.grad
                    if gr.dtype != torch.float32 or not gr.is_contiguous() or not p.is_contiguous():
                        raise RuntimeError("FusedSGD expects contiguous float32 parameters and gradients")
                    r.param, r.grad, r.count, r.flat_offset = p.data_ptr(), gr.data_ptr(), p.numel(), self._rest_off[i]
                    r.first_step = int(i not in self._rest_started)      # mom == 0: the buffer is written but never read back
                _lib.check(L.mmnn_sgd_step_multi(refs, len(chunk), self._rest_buf.data_ptr(), lr, mom, wd, nes, st), "sgd_step_multi")
            self._rest_started.update(i for i, _ in live)
        return None

    def zero_grad(self, set_to_none: bool = True):
        """Tail gradients are set to None.  The backbone's `.grad` views stay attached to the flat gradient buffer and are marked
        stale: the next backward OVERWRITES the buffer (so accumulation semantics are those of zero_grad), without touching 364
        tensors per step.  Until that backward the stale values remain readable through `.grad`."""
        for bb in self._backbones:
            bb.mark_grads_stale()
        for p in self._rest:
            p.grad = None
