"""MI355X-native mirror of the clinical-feature MLP (models/mlp.py:7-63 of DigITs-AIML/MMNN_STS).

Same constructor, attributes (`backbone`, `features`, `output_head`) and state_dict keys.  Each Sequential runs as ONE
fused HIP kernel (Linear -> BatchNorm1d -> ReLU/Dropout1d chain); the nn.Linear / nn.BatchNorm1d children only hold
the parameters.  Reproduced reference behaviour:
  * layer 0 is dense -> bn -> relu -> drop, layers 1..5 are dense -> bn -> drop -> relu;
  * `nn.Linear(32, 16, 3)`: the third positional argument is `bias` (truthy) -- the layers simply have a bias;
  * `nn.Dropout1d` on a 2-D (N, F) input treats it as un-batched (C=N, L=F): whole ROWS (patients) are zeroed.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops

_WIDTHS = (32, 16, 8, 8, 8)


class _Stack(nn.Sequential):
    """A run of dense{i}/bn{i}/relu{i}/drop{i} groups executed by one kernel."""

    def __init__(self, modules: "OrderedDict[str, nn.Module]", indices, relu_first, dropout_prob, first_id):
        super().__init__(modules)
        self._idx, self._relu_first, self._p, self._first_id = tuple(indices), tuple(relu_first), float(dropout_prob), first_id

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        params, running, bns = [], [], []
        for i in self._idx:
            d, b = getattr(self, f"dense{i}"), getattr(self, f"bn{i}")
            params += [d.weight, d.bias, b.weight, b.bias]
            running += [b.running_mean, b.running_var]
            bns.append(b)
        p = float(getattr(self, f"drop{self._idx[0]}").p)
        if bns[0].momentum is None:
            raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative moving average) is not available in the fused MLP stack; "
                                      "the reference builds its norms with the default momentum (models/mlp.py:22-48)")
        counters = [b.num_batches_tracked for b in bns] if x.is_cuda else None      # incremented by the forward kernel itself
        cfg = (self._relu_first, p, bns[0].eps, bns[0].momentum, self.training, self._first_id, counters)
        return ops.MlpStack.apply(x, cfg, running, *params)


class _Head(nn.Sequential):
    def forward(self, f: torch.Tensor) -> torch.Tensor:
        return ops.SmallLinear.apply(f, self.dense6.weight, self.dense6.bias)


class MLP(nn.Module):
    def __init__(self, in_channels=1, out_channels=3, feature_channels=12, dropout_prob=0.2):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.feature_channels, self.dropout_prob = feature_channels, dropout_prob
        self.relu = nn.ReLU()
        mods = OrderedDict()
        prev = in_channels
        for i, w in enumerate(_WIDTHS):
            mods[f"dense{i}"] = nn.Linear(prev, w)
            mods[f"bn{i}"] = nn.BatchNorm1d(w)
            if i == 0:
                mods["relu0"], mods["drop0"] = self.relu, nn.Dropout1d(dropout_prob)
            else:
                mods[f"drop{i}"], mods[f"relu{i}"] = nn.Dropout1d(dropout_prob), self.relu
            prev = w
        self.backbone = _Stack(mods, range(5), [True, False, False, False, False], dropout_prob, 0)
        self.features = _Stack(OrderedDict([
            ("dense5", nn.Linear(prev, feature_channels)), ("bn5", nn.BatchNorm1d(feature_channels)),
            ("drop5", nn.Dropout1d(dropout_prob)), ("relu5", self.relu)]), [5], [False], dropout_prob, 5)
        self.output_head = _Head(OrderedDict([("dense6", nn.Linear(feature_channels, out_channels))]))

    def forward(self, x):
        return self.output_head(self.features(self.backbone(x)))
