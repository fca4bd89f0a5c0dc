"""Mirror of the helpers of utils/utils.py (DigITs-AIML/MMNN_STS) that sit on the fusion training path.

Kept (same names / signatures): `criterion` (:20-22), `surv_criterion` (:24-29), `save_model` (:31-35),
`BackpropagatableFeatureExtractor` (:238-251), `MultiModalGradCAM` (:253-344), `loadWeights` (:357-390, local files),
`add_gradcam` (:451-455), `loadUIDs` (:175-181, local files).  S3 / matplotlib / medcam helpers are host-side I/O
outside the hot path and are not reproduced.
"""
import logging
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, ops

logger = logging.getLogger(__name__)


def criterion(loss_func, preds, labels, device):
    return loss_func(preds, labels).to(device)


def surv_criterion(loss_func, preds, events, durations, device):
    """Sum of the survival loss over the C target columns (utils/utils.py:24-29).  With the package's `CoxPH` on the GPU all
    columns are evaluated by one kernel; any other callable is applied column by column like the reference does."""
    from ..losses import losses as _l
    if loss_func is _l.CoxPH and preds.is_cuda and preds.dim() == 2:
        loss, _ = ops.CoxBlend.apply(preds.unsqueeze(0), events, durations, None)
        return loss
    total = 0
    for i in range(preds.shape[1]):
        total = total + loss_func(preds[:, i], events[:, i], durations[:, i]).to(device)
    return total


def save_model(model, model_dir):
    logger.info("Saving the model.")
    torch.save(model.cpu().state_dict(), os.path.join(model_dir, 'model.pth'))


def loadUIDs(path):
    with open(path) as f:
        return [int(line.strip()) for line in f.readlines()]


class BackpropagatableFeatureExtractor(nn.Module):
    """`features(backbone(x))`, skipping the wrapped model's own classifier (utils/utils.py:238-251)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, x):
        return self.model.features(self.model.backbone(x))


class MultiModalGradCAM(nn.Module):
    """Grad-CAM of the fusion model on the LAST Conv3d of the image backbone (= last dense layer's conv2), reproducing
    utils/utils.py:253-344 including its quirks: batch size 1 only (:334), channel-pooled gradients weight the activations
    IN PLACE and CUMULATIVELY across classes (:313-314), min-max normalisation, trilinear up-sampling to the input size.

    The reference runs a full autograd backward per class and keeps only d out[0,cls] / d act.  Here the eval-mode forward
    is the HIP backbone and that single gradient is evaluated in closed form (head -> feature_layer -> GAP -> ReLU mask ->
    norm5 scale, restricted to the last `growth_rate` channels) -- no weight gradients are computed.
    """

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.input_shape = None
        self.features = None
        self.grads = None

    def forward(self, x):
        mm = self.model
        if getattr(mm, "blend", False):
            raise ValueError("MultiModalGradCAM expects blend=False (a (N, C) output), as main.py's inference path uses it")
        dn = mm.image_model.model
        bb = dn.backbone
        was_training = mm.training
        mm.eval()
        with torch.no_grad():
            image = x['image']
            assert image.shape[0] == 1, 'Batch dimension found in attention map - Must use batch size 1 when computing attention maps'
            h = bb(image)                                           # norm5 output, eval statistics
            ent = bb._plans[(tuple(image.shape), image.device.index)]
            L = _lib.lib()
            nb = len(bb.cfg["block_config"]) - 1
            off = L.mmnn_densenet_ws_offset(ent["plan"], b"x", nb, 0)
            ctot, g = h.shape[1], bb.cfg["growth_rate"]
            sp = tuple(h.shape[2:])
            v = h[0, 0].numel()
            xlast = ent["ws"][off:off + 4 * ctot * v].view(torch.float32).view(1, ctot, *sp)
            act = xlast[:, ctot - g:].clone()                        # output of the last conv2 (dropout is off in eval)
            fi = dn.features(h)
            fc = mm.clinical_model(x['clinical'])
            outputs = ops.FusionHeads.apply(fi, fc, mm.output_head.weight, mm.output_head.bias, mm.image_output_head.weight,
                                            mm.image_output_head.bias, mm.clinical_output_head.weight, mm.clinical_output_head.bias, False)
            # d out[0, cls] / d act[c, v] = sum_f Wout[cls, f] * Wfeat[f, c'] / V * [h[c', v] > 0] * a5[c'],  c' = ctot - g + c
            n5 = bb.norm5
            a5 = (n5.weight / torch.sqrt(n5.running_var + n5.eps))[ctot - g:]
            wfeat = dn.features.feature_layer.weight[:, ctot - g:]                     # (F, g)
            wout = mm.output_head.weight[:, :wfeat.shape[0]]                           # (C, F) image half of the fused head
            chan = (wout @ wfeat) * a5 / float(v)                                      # (C, g)
            mask = (h[:, ctot - g:] > 0).to(torch.float32)                             # (1, g, d, h, w)
            self.input_shape = image.shape
            att_maps = []
            for cls in range(outputs.shape[1]):
                grads = mask * chan[cls].view(1, -1, 1, 1, 1)
                self.grads = grads
                pooled = grads.mean(dim=[0, 2, 3, 4])
                act *= pooled.view(1, -1, 1, 1, 1)                                      # in place, cumulative (reference quirk)
                heat = act.mean(dim=1).squeeze()
                heat = heat - heat.min()
                heat = heat / heat.max()
                assert heat.ndim == 3, 'Batch dimension found in attention map - Must use batch size 1 when computing attention maps'
                att_maps.append(F.interpolate(heat[None, None], self.input_shape[2:], mode='trilinear').squeeze())
            self.features = act
        mm.train(was_training)
        return outputs, att_maps


def add_gradcam(model, output_dir='attention_maps', multimodal=False):
    if multimodal:
        return model.add_gradcam(output_dir)
    raise NotImplementedError("unimodal Grad-CAM in the reference is the third-party `medcam` package (utils/utils.py:455); "
                              "only the multimodal Grad-CAM is part of this path")


def loadWeights(model, path, device):
    """utils/utils.py:357-390 for local files: plain state_dict, or the BHB-10K pretrained DenseNet121 key remap."""
    checkpoint = torch.load(path, map_location=device)
    if isinstance(checkpoint, dict) and 'model' in checkpoint and path.endswith('DenseNet121_BHB-10K_yAwareContrastive.pth'):
        remapped = {}
        for key, value in checkpoint['model'].items():
            parts = key.replace('module.', '').split('.')
            if parts[0] == 'features' and parts[1].startswith('dense'):
                parts.insert(3, 'layers')
            remapped['.'.join(parts)] = value
        model.load_state_dict(remapped, strict=False)
        logger.info('Loaded pretrained backbone')
    else:
        model.load_state_dict(checkpoint)
        logger.info('Loaded provided weights from disk')
    return model
