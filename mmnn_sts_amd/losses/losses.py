"""Cox partial-likelihood loss of the reference (losses/losses.py:6-9), on the MI355X.

The reference calls `pycox.models.loss.CoxPHLoss()(log_h, events, duration)`, i.e. it forwards its arguments POSITIONALLY
into pycox's `(log_h, durations, events)`: the event flags end up as the sort key and the durations as the weights
(SURVEY 8(a) A9).  That call is reproduced verbatim; `intended_order=True` gives pycox's documented semantics.
The sort is a STABLE descending sort (torch.sort's tie order is unspecified for n > 16).
"""
import torch

from .. import ops

NUM_CLASSES = 2


def cox_ph_loss(log_h: torch.Tensor, durations: torch.Tensor, events: torch.Tensor) -> torch.Tensor:
    """pycox `CoxPHLoss.forward(log_h, durations, events)` for one target vector."""
    loss, _ = ops.CoxBlend.apply(log_h.reshape(1, -1, 1), durations.reshape(-1, 1), events.reshape(-1, 1), None)
    return loss


def CoxPH(log_h, events, duration, intended_order: bool = False):
    if intended_order:
        return cox_ph_loss(log_h, duration, events)
    return cox_ph_loss(log_h, events, duration)


class BCEWithLogitsLoss(torch.nn.Module):
    """`nn.BCEWithLogitsLoss(pos_weight=..., reduction=...)` as the classification trainer builds it (main.py:147-153), on the
    MI355X: the element-wise loss and its derivative come from one HIP kernel."""

    def __init__(self, pos_weight=None, reduction: str = 'mean'):
        super().__init__()
        if reduction not in ('none', 'sum', 'mean'):
            raise ValueError(f"{reduction} is not a valid value for reduction")
        self.register_buffer('pos_weight', pos_weight)
        self.reduction = reduction

    def forward(self, input, target):
        if target.shape != input.shape:
            raise ValueError(f"Target size ({target.shape}) must be the same as input size ({input.shape})")
        loss = ops.BceLogits.apply(input, target.to(input.dtype), self.pos_weight)
        return loss if self.reduction == 'none' else (loss.sum() if self.reduction == 'sum' else loss.mean())
