"""Gradient blending for the fusion model's three heads (interface of losses/GradientBlender.py:9-257 of DigITs-AIML/MMNN_STS;
method of Wang et al., arXiv:1905.12681).

The class keeps the reference's public surface -- constructor arguments, `computeLoss` / `updateWeights` / `saveHistory` and
their survival / classification variants, the attributes `weights`, `history`, `lvn`, `ltn`, `reduction`, `survival`,
`surv_criterion` -- because callers (main.py:417,464,584-590) use exactly those.  The implementation is this package's own:

* survival (`computeLossSurv`, `updateWeightsSurv`): when the loss is the package's `CoxPH`, all heads x targets Cox losses,
  their blend and the gradient come from ONE HIP kernel (`ops.CoxBlend`); any other callable is evaluated head by head
  through `surv_criterion` like upstream (:197).
* both weight updates go through `_reweigh`, which differs between the two tasks only in the sign convention of the
  generalisation term (upstream :91 vs :128 -- reproduced), and normalises with a softmax over the heads
  (:247-253 uses an implicit-dim softmax; dim 0 here).
* bookkeeping quirks that change observable state are kept: the survival path records weights in `history` only on
  `updateWeights` (:103), the classification path also on its first loss (:168-170) and NOT on its updates (:134-136).
* fixed: weights follow the device of the predictions (upstream mixes CPU weights with GPU losses, Appendix A Q4).
"""
import numpy as np
import torch

from .. import ops
from . import losses as _losses

_BAD_REDUCTION = 'Unable to reduce loss, unrecognized reduction: {}'

# reduction name prefix -> (reduce everything, reduce all but the head axis)
_REDUCERS = (
    ('sum', torch.sum, lambda t: t.sum(dim=(1, 2))),
    ('mean', torch.mean, lambda t: t.mean(dim=(1, 2))),
    ('none', lambda t: t, lambda t: t),
)


class GradientBlender:
    def __init__(self, loss_function, survival=False, reduction='sum', device='cpu', surv_criterion=None):
        self.loss_function, self.surv_criterion = loss_function, surv_criterion
        self.survival, self.device = survival, device
        self.reduction = reduction.lower()
        self.weights = self.lvn = self.ltn = None
        self.history = []

    # ---- shared machinery ------------------------------------------------------------------------------------------------
    def _reducer(self, per_head: bool):
        if self.reduction is None:
            return _REDUCERS[2][1]
        for prefix, full, heads in _REDUCERS:
            if self.reduction.startswith(prefix):
                return heads if per_head else full
        raise ValueError(_BAD_REDUCTION.format(self.reduction))

    def reduce(self, loss):
        """Collapse a loss tensor of any shape according to `reduction` ('none': unchanged)."""
        return self._reducer(False)(loss)

    def reduceToHeads(self, loss):
        """(heads, N, C) -> (heads,) according to `reduction` ('none': unchanged)."""
        return self._reducer(True)(loss)

    def normalize(self, weights):
        return torch.softmax(weights, dim=0)

    def _uniform(self, heads: int, device=None):
        w = self.normalize(torch.ones(heads))
        return w if device is None else w.to(device)

    def _record(self):
        self.history.append(self.weights.detach().cpu().numpy())

    def _reweigh(self, train_loss, val_loss, gain_sign: float):
        """One G/O^2 step from the per-head losses of two checkpoints.  O = val - train (overfitting) at the previous and the
        current checkpoint, G = gain_sign * (previous val - current val); weights = softmax(G / dO^2).  The first call only
        stores the checkpoint and resets the weights to uniform."""
        dev = train_loss.device
        if self.lvn is None or self.ltn is None:
            self.weights = self._uniform(train_loss.shape[0], dev)
        else:
            gain = gain_sign * (self.lvn - val_loss)
            overfit_growth = (val_loss - train_loss) - (self.lvn - self.ltn)
            self.weights = self.normalize(gain / overfit_growth.square()).to(dev)
        self.lvn, self.ltn = val_loss, train_loss

    def updateWeights(self, *args, **kwargs):
        return (self.updateWeightsSurv if self.survival else self.updateWeightsClass)(*args, **kwargs)

    def computeLoss(self, *args, **kwargs):
        return (self.computeLossSurv if self.survival else self.computeLossClassification)(*args, **kwargs)

    def saveHistory(self):
        np.savetxt('gblend_weights_history.csv', np.array(self.history), delimiter=',')

    # ---- survival ----------------------------------------------------------------------------------------------------------
    def _head_losses(self, preds, events, durations, weights=None):
        """(sum_h weights[h] * L[h] or None, L) with L[h] = surv_criterion(loss, preds[h], events, durations)."""
        if self.loss_function is _losses.CoxPH and preds.is_cuda:
            return ops.CoxBlend.apply(preds, events, durations, weights)
        per_head = torch.stack([self.surv_criterion(self.loss_function, p, events, durations, preds.device) for p in preds.unbind(0)])
        return (None if weights is None else (weights * per_head).sum()), per_head

    def computeLossSurv(self, preds, events, durations, reduceToHeads=False):
        """preds (heads, N, C); events / durations (N, C).  Returns (blended loss, loss of head 0), or the per-head losses."""
        if self.weights is None:
            self.weights = self._uniform(preds.shape[0])
        if reduceToHeads:
            return self._head_losses(preds, events, durations)[1]
        w = self.weights = self.weights.to(preds.device)
        if self.reduction is not None and self.reduction.startswith('sum'):
            blended, per_head = self._head_losses(preds, events, durations, w)
        elif self.reduction is not None and self.reduction.startswith('mean'):
            blended, per_head = self._head_losses(preds, events, durations, w / preds.shape[0])
        else:
            per_head = self._head_losses(preds, events, durations)[1]
            blended = self.reduce(w * per_head)        # 'none' -> per-head vector; anything else raises ValueError
        return blended, per_head[0]

    def updateWeightsSurv(self, train_preds, train_events, train_durations, val_preds, val_events, val_durations):
        dev = train_preds.device
        with torch.no_grad():
            tl = self.computeLossSurv(train_preds, train_events.to(dev), train_durations.to(dev), reduceToHeads=True)
            vl = self.computeLossSurv(val_preds.to(dev), val_events.to(dev), val_durations.to(dev), reduceToHeads=True)
        self._reweigh(tl, vl, +1.0)
        self._record()

    # ---- classification ------------------------------------------------------------------------------------------------------
    def computeLossClassification(self, preds, targets, reduceToHeads=False, no_reduce=False):
        """preds (heads, N, C), targets (N, C): the element-wise loss of every head against the same targets, then reduced
        per head and blended.  `no_reduce`: the raw (heads, N, C) tensor; `reduceToHeads`: the per-head losses."""
        raw = self.loss_function(preds, targets.unsqueeze(0).expand(preds.shape[0], *targets.shape))
        if self.weights is None:
            self.weights = self._uniform(preds.shape[0])
            self._record()
        if no_reduce:
            return raw
        per_head = self.reduceToHeads(raw)
        if reduceToHeads:
            return per_head
        self.weights = self.weights.to(per_head.device)
        return self.reduce(self.weights * per_head)

    def updateWeightsClass(self, train_preds, train_targs, val_preds, val_targs):
        dev = train_preds.device
        with torch.no_grad():
            tl = self.computeLossClassification(train_preds, train_targs.to(dev), reduceToHeads=True)
            vl = self.computeLossClassification(val_preds.to(dev), val_targs.to(dev), reduceToHeads=True)
        self._reweigh(tl, vl, -1.0)
