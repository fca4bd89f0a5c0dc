"""Fused SGD (momentum / Nesterov / weight decay) for the fusion model -- the optimizer of main.py:410-413.

torch.optim.SGD works unchanged on the native modules (their parameters are ordinary nn.Parameters); this class does the
same update with ONE HIP launch over the backbone's flat parameter buffer (11.26 M of the 11.28 M parameters) plus a few
foreach ops for the ~30 small tail tensors.  It is a torch.optim.Optimizer, so lr schedulers (OneCycleLR, which also
cycles `momentum`) drive it through `param_groups` as usual.
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
        self._rest_bufs = None

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        lr, mom, wd, nes = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), int(bool(g["nesterov"]))
        st = torch.cuda.current_stream().cuda_stream
        for bb in self._backbones:
            flat, grad = bb.flat_parameters, bb.flat_grad
            if grad is None or bb.conv0.weight.grad is None or bb._grads_stale:
                continue          # no gradient since the last zero_grad()
            first = id(bb) not in self._bufs or self._bufs[id(bb)].data_ptr() == 0 or self._bufs[id(bb)].numel() != flat.numel()
            if first:
                self._bufs[id(bb)] = torch.empty_like(flat)
            _lib.check(_lib.lib().mmnn_sgd_step(flat.data_ptr(), grad.data_ptr(), self._bufs[id(bb)].data_ptr(), flat.numel(), lr, mom, wd,
                                                nes, int(first), st), "sgd_step")
            bb.mark_params_changed()          # written through the raw pointer: invisible to autograd's version counters
        ps = [p for p in self._rest if p.grad is not None]
        if ps:
            gs = [p.grad for p in ps]
            if wd != 0.0:
                gs = torch._foreach_add(gs, ps, alpha=wd)
            if mom != 0.0:
                if self._rest_bufs is None or len(self._rest_bufs) != len(ps):
                    self._rest_bufs = [t.clone() for t in gs]
                else:
                    torch._foreach_mul_(self._rest_bufs, mom)
                    torch._foreach_add_(self._rest_bufs, gs)
                gs = torch._foreach_add(gs, self._rest_bufs, alpha=mom) if nes else self._rest_bufs
            torch._foreach_add_(ps, gs, alpha=-lr)
        return None

    def zero_grad(self, set_to_none: bool = True):
        """Tail gradients are set to None.  The backbone's `.grad` views stay attached to the flat gradient buffer and are marked
        stale: the next backward OVERWRITES the buffer (so accumulation semantics are those of zero_grad), without touching 364
        tensors per step.  Until that backward the stale values remain readable through `.grad`."""
        for bb in self._backbones:
            bb.mark_grads_stale()
        for p in self._rest:
            p.grad = None
