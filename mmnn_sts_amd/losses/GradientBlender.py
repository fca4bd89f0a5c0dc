"""Gradient blending (losses/GradientBlender.py:9-257 of DigITs-AIML/MMNN_STS; Wang et al., arXiv:1905.12681).

Same class name, constructor and methods.  The survival branch (`computeLossSurv`, `updateWeightsSurv`) evaluates all
heads x targets Cox losses and their blend in ONE HIP kernel when the loss is the package's `CoxPH`; any other loss
callable goes through `surv_criterion` head by head exactly like the reference.  Reference quirks kept: softmax
normalisation (:247-253, dim 0), dG sign per branch (:91 vs :128), history only appended by `updateWeights` for
survival (:103) but also by the first classification loss (:168-170).  Fixed: weights follow the predictions' device
(the reference mixes CPU weights with GPU losses).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from . import losses as _losses


class GradientBlender:
    def __init__(self, loss_function, survival=False, reduction='sum', device='cpu', surv_criterion=None):
        self.loss_function = loss_function
        self.weights = None
        self.reduction = reduction.lower()
        self.survival = survival
        self.lvn = None
        self.ltn = None
        self.device = device
        self.surv_criterion = surv_criterion
        self.history = []

    # ---- survival ----------------------------------------------------------------------------------------------------
    def _native(self) -> bool:
        return self.loss_function is _losses.CoxPH

    def _head_losses(self, preds, events, durations, weights=None):
        if self._native() and preds.is_cuda:
            return ops.CoxBlend.apply(preds, events, durations, weights)     # (blend, head_losses)
        hl = torch.stack([self.surv_criterion(self.loss_function, preds[i, ...], events, durations, preds.device)
                          for i in range(preds.shape[0])], dim=0)
        return (None if weights is None else torch.sum(weights * hl)), hl

    def computeLossSurv(self, preds, events, durations, reduceToHeads=False):
        if self.weights is None:
            self.weights = self.normalize(torch.ones(preds.shape[0]))
        if reduceToHeads:
            return self._head_losses(preds, events, durations)[1]
        self.weights = self.weights.to(preds.device)
        if self.reduction.startswith('sum'):
            blended, hl = self._head_losses(preds, events, durations, self.weights)
        elif self.reduction.startswith('mean'):
            blended, hl = self._head_losses(preds, events, durations, self.weights / preds.shape[0])
        elif self.reduction.startswith('none'):
            hl = self._head_losses(preds, events, durations)[1]
            blended = self.weights * hl
        else:
            raise ValueError('Unable to reduce loss, unrecognized reduction: {}'.format(self.reduction))
        return blended, hl[0]

    def updateWeightsSurv(self, train_preds, train_events, train_durations, val_preds, val_events, val_durations):
        dev = train_preds.device
        with torch.no_grad():
            train_loss = self.computeLossSurv(train_preds, train_events.to(dev), train_durations.to(dev), reduceToHeads=True)
            val_loss = self.computeLossSurv(val_preds.to(dev), val_events.to(dev), val_durations.to(dev), reduceToHeads=True)
        if self.lvn is None or self.ltn is None:
            self.weights = self.normalize(torch.ones(train_preds.shape[0])).to(dev)
        else:
            o_n = self.lvn - self.ltn
            o_npn = val_loss - train_loss
            delta_g = self.lvn - val_loss
            delta_o = o_npn - o_n
            self.weights = self.normalize(delta_g / torch.pow(delta_o, 2)).to(dev)
        self.lvn = val_loss
        self.ltn = train_loss
        self.history.append(self.weights.detach().cpu().numpy())

    # ---- classification (losses/GradientBlender.py:105-136,150-179) --------------------------------------------------------
    def updateWeightsClass(self, train_preds, train_targs, val_preds, val_targs):
        dev = train_preds.device
        train_loss = self.computeLossClassification(train_preds, train_targs.to(dev), reduceToHeads=True)
        val_loss = self.computeLossClassification(val_preds.to(dev), val_targs.to(dev), reduceToHeads=True)
        if self.lvn is None or self.ltn is None:
            self.weights = self.normalize(torch.ones(train_preds.shape[0])).to(dev)
        else:
            o_n = self.lvn - self.ltn
            o_npn = val_loss - train_loss
            delta_g = val_loss - self.lvn
            delta_o = o_npn - o_n
            self.weights = self.normalize(delta_g / torch.pow(delta_o, 2)).to(dev)
        self.lvn = val_loss
        self.ltn = train_loss

    def computeLossClassification(self, preds, targets, reduceToHeads=False, no_reduce=False):
        targets = torch.stack([targets for _ in range(preds.shape[0])], dim=0)
        loss = self.loss_function(preds, targets)
        if self.weights is None:
            self.weights = self.normalize(torch.ones(preds.shape[0]))
            self.history.append(self.weights.detach().cpu().numpy())
        if no_reduce:
            return loss
        loss = self.reduceToHeads(loss)
        if reduceToHeads:
            return loss
        self.weights = self.weights.to(device=loss.device)
        return self.reduce(self.weights * loss)

    # ---- dispatch / helpers ---------------------------------------------------------------------------------------------
    def updateWeights(self, *args, **kwargs):
        if self.survival:
            self.updateWeightsSurv(*args, **kwargs)
        else:
            self.updateWeightsClass(*args, **kwargs)

    def computeLoss(self, *args, **kwargs):
        if self.survival:
            return self.computeLossSurv(*args, **kwargs)
        return self.computeLossClassification(*args, **kwargs)

    def reduceToHeads(self, loss):
        if self.reduction.startswith('sum'):
            return torch.sum(loss, dim=(1, 2))
        elif self.reduction.startswith('mean'):
            return torch.mean(loss, dim=(1, 2))
        elif self.reduction.startswith('none') or self.reduction is None:
            return loss
        raise ValueError('Unable to reduce loss, unrecognized reduction: {}'.format(self.reduction))

    def reduce(self, loss):
        if self.reduction.startswith('sum'):
            return torch.sum(loss)
        elif self.reduction.startswith('mean'):
            return torch.mean(loss)
        elif self.reduction.startswith('none') or self.reduction is None:
            return loss
        raise ValueError('Unable to reduce loss, unrecognized reduction: {}'.format(self.reduction))

    def normalize(self, weights):
        return F.softmax(weights, dim=0)

    def saveHistory(self):
        np.savetxt('gblend_weights_history.csv', np.array(self.history), delimiter=',')
