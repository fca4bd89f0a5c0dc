"""MI355X-native mirror of the reference's 3-D ResNet-18 variant (models/resnet.py:5-227 of DigITs-AIML/MMNN_STS).

Same class names, constructor arguments and `state_dict` schema (`stem.0/1`, `layer{1..4}.{i}.conv1.0/1`, `.conv2.0/1`,
`.downsample.0/1`, `fc`), so checkpoints are interchangeable.  The torch.nn children are parameter containers; every
`forward` below runs HIP kernels through the C-ABI (include/mmnn_sts.h: mmnn_conv3d_*, mmnn_bn3d_*, mmnn_gap_fc_sigmoid_*):
direct convolutions (the net is 8 / 16 channels wide), BatchNorm fused with the residual add, the ReLU and the per-stage
dropout, pooled sigmoid head.  Upstream this encoder is reachable only standalone (parser/parser.py:151-160: it has no
`.backbone` / `.features`, so `MultiModalModel` rejects it); the same holds here.
"""
import torch
import torch.nn as nn

from .. import ops

__all__ = ["BasicStem", "BasicBlock", "Bottleneck", "Conv3DSimple", "Resnet18", "r3d_18"]


def _conv(x, conv: nn.Conv3d):
    return ops.Conv3dDirect.apply(x, conv.weight, conv.stride, conv.padding)


def _bn(x, bn: nn.BatchNorm3d, relu: bool, residual=None, drop_p: float = 0.0):
    training = bn.training
    # momentum=None is torch's cumulative moving average: factor 1 / (batches seen including this one)
    momentum = bn.momentum if bn.momentum is not None else 1.0 / (int(bn.num_batches_tracked) + 1)
    out = ops.BatchNormAct3d.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, momentum,
                                   bn.eps, training, relu, drop_p if training else 0.0)
    if training:
        bn.num_batches_tracked.add_(1)
    return out


class BasicStem(nn.Sequential):
    """conv (1,7,7) stride (1,2,2) padding (1,3,3) -> BN -> ReLU  (models/resnet.py:5-13)."""

    def __init__(self):
        super().__init__(nn.Conv3d(1, 64, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False), nn.BatchNorm3d(64),
                         nn.ReLU(inplace=True))

    def forward(self, x):
        return _bn(_conv(x, self[0]), self[1], relu=True)


class Conv3DSimple(nn.Conv3d):
    """models/resnet.py:95-112."""

    def __init__(self, in_planes: int, out_planes: int, midplanes=None, stride: int = 1, padding: int = 1) -> None:
        super().__init__(in_channels=in_planes, out_channels=out_planes, kernel_size=(3, 3, 3), stride=stride, padding=padding, bias=False)

    @staticmethod
    def get_downsample_stride(stride: int):
        return stride, stride, stride


def _midplanes(inplanes, planes):
    return (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)


class BasicBlock(nn.Module):
    """models/resnet.py:60-93: relu(bn2(conv2(relu(bn1(conv1(x))))) + residual)."""
    expansion = 1

    def __init__(self, inplanes: int, planes: int, conv_builder, stride: int = 1, downsample=None) -> None:
        super().__init__()
        mid = _midplanes(inplanes, planes)
        self.conv1 = nn.Sequential(conv_builder(inplanes, planes, mid, stride), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(conv_builder(planes, planes, mid), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, drop_p: float = 0.0):
        out = _bn(_conv(x, self.conv1[0]), self.conv1[1], relu=True)
        out = _conv(out, self.conv2[0])
        residual = x if self.downsample is None else _bn(_conv(x, self.downsample[0]), self.downsample[1], relu=False)
        return _bn(out, self.conv2[1], relu=True, residual=residual, drop_p=drop_p)


class Bottleneck(nn.Module):
    """models/resnet.py:15-58 (not used by r3d_18; same kernels)."""
    expansion = 4

    def __init__(self, inplanes, planes, conv_builder, stride=1, downsample=None):
        super().__init__()
        mid = _midplanes(inplanes, planes)
        self.conv1 = nn.Sequential(nn.Conv3d(inplanes, planes, kernel_size=1, bias=False), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(conv_builder(planes, planes, mid, stride), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(nn.Conv3d(planes, planes * self.expansion, kernel_size=1, bias=False), nn.BatchNorm3d(planes * self.expansion))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, drop_p: float = 0.0):
        out = _bn(_conv(x, self.conv1[0]), self.conv1[1], relu=True)
        out = _bn(_conv(out, self.conv2[0]), self.conv2[1], relu=True)
        out = _conv(out, self.conv3[0])
        residual = x if self.downsample is None else _bn(_conv(x, self.downsample[0]), self.downsample[1], relu=False)
        return _bn(out, self.conv3[1], relu=True, residual=residual, drop_p=drop_p)


# (planes, stride) of layer1 .. layer4 (models/resnet.py:134-137) and the width the stem hands to layer1 (:129)
_STAGE_PLAN = ((8, 1), (16, 2), (8, 2), (16, 2))
_STEM_WIDTH = 64


def _build_stage(block, conv_builder, width_in: int, planes: int, depth: int, stride: int):
    """One residual stage: `depth` blocks, the first of which may change stride / width and then owns a projection shortcut
    (1x1x1 conv + BN, models/resnet.py:170-181).  Returns (stage, width_out)."""
    width_out = planes * block.expansion
    shortcut = None
    if stride != 1 or width_in != width_out:
        shortcut = nn.Sequential(nn.Conv3d(width_in, width_out, kernel_size=1, stride=conv_builder.get_downsample_stride(stride), bias=False),
                                 nn.BatchNorm3d(width_out))
    widths = [width_in] + [width_out] * (depth - 1)
    blocks = [block(w, planes, conv_builder, stride, shortcut) if k == 0 else block(w, planes, conv_builder) for k, w in enumerate(widths)]
    return nn.Sequential(*blocks), width_out


def _init_module(m: nn.Module) -> None:
    """Initial values per module type (models/resnet.py:186-199)."""
    if isinstance(m, nn.Conv3d):
        nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm3d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, 0, 0.01)
        nn.init.zeros_(m.bias)


class Resnet18(nn.Module):
    """models/resnet.py:114-199: stem, four stages of `block`s (planes 8 / 16 / 8 / 16, strides 1 / 2 / 2 / 2), element-wise
    dropout after every stage, global average pool, Linear, sigmoid.  Table-driven: `_STAGE_PLAN` x (conv_makers, layers)."""

    def __init__(self, block, conv_makers, layers, stem, num_classes=400, zero_init_residual=False, dropout_prob=0.2):
        super().__init__()
        self.stem = stem()
        self.dropout = nn.Dropout(p=dropout_prob)
        width = _STEM_WIDTH
        for k, ((planes, stride), maker, depth) in enumerate(zip(_STAGE_PLAN, conv_makers, layers), start=1):
            stage, width = _build_stage(block, maker, width, planes, depth, stride)
            self.add_module(f"layer{k}", stage)
        self.inplanes = width                                     # attribute kept for callers that read it
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(width, num_classes)
        self.apply(_init_module)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.conv3[1].weight)             # upstream names a non-existent `bn3` here (:147-150)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mmnn_sts_amd: r3d_18 runs on the MI355X only (no CPU path); move model and input to cuda")
        if x.dim() != 5 or x.shape[1] != self.stem[0].in_channels:
            raise ValueError(f"expected (N, {self.stem[0].in_channels}, D, H, W) input, got {tuple(x.shape)}")
        x = self.stem(x)
        p = float(self.dropout.p) if self.training else 0.0
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for i, blk in enumerate(stage):
                x = blk(x, drop_p=p if i == len(stage) - 1 else 0.0)     # self.dropout(x) after each stage (:159-166), fused
        return ops.GapFcSigmoid.apply(x, self.fc.weight, self.fc.bias)


def r3d_18(num_classes):
    """models/resnet.py:202-227."""
    return Resnet18(BasicBlock, [Conv3DSimple] * 4, [2, 2, 2, 2], BasicStem, num_classes=num_classes)
