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

    The reference runs a full autograd backward per class and keeps only d out[0,cls] / d act.  Here the eval-mode forward is the
    HIP backbone and everything after it is ONE C-ABI call (`mmnn_gradcam`, csrc/gradcam.hip): that single gradient in closed form
    (head -> feature_layer -> GAP -> ReLU mask -> norm5 scale, restricted to the last `growth_rate` channels; no weight gradients),
    the pooled weighting, the channel mean, the min-max normalisation and the trilinear up-sampling.  torch allocates the outputs.
    """

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.input_shape = None
        self.features = None
        self.grads = None

    def forward(self, x):
        import ctypes
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
            ctot, g = h.shape[1], bb.cfg["growth_rate"]
            sp = tuple(h.shape[2:])
            v = h[0, 0].numel()
            # output of the last conv2 (dropout is off in eval) = the last `g` channels of the last block's concat buffer
            act_in = ent["ws"].data_ptr() + L.mmnn_densenet_ws_offset(ent["plan"], b"x", nb, 0) + 4 * (ctot - g) * v
            fi = dn.features(h)
            fc = mm.clinical_model(x['clinical'])
            outputs = ops.FusionHeads.apply(fi, fc, mm.output_head.weight, mm.output_head.bias, mm.image_output_head.weight,
                                            mm.image_output_head.bias, mm.clinical_output_head.weight, mm.clinical_output_head.bias, False)
            n5, wfeat, whead = bb.norm5, dn.features.feature_layer.weight, mm.output_head.weight
            ncls = outputs.shape[1]
            dev = h.device
            self.input_shape = image.shape
            D, H, W = (int(t) for t in image.shape[2:])
            act = torch.empty((1, g) + sp, device=dev, dtype=torch.float32)
            grads = torch.empty((1, g) + sp, device=dev, dtype=torch.float32)
            heat = torch.empty((ncls,) + sp, device=dev, dtype=torch.float32)
            maps = torch.empty((ncls, D, H, W), device=dev, dtype=torch.float32)
            desc = _lib.GradcamDesc(ctot, g, sp[0], sp[1], sp[2], ncls, wfeat.shape[0], whead.shape[1], D, H, W, float(n5.eps))
            for t in (h, wfeat, whead, n5.weight, n5.running_var):
                if not (t.is_contiguous() and t.dtype == torch.float32):
                    raise RuntimeError("MultiModalGradCAM expects contiguous float32 model tensors")
            _lib.check(L.mmnn_gradcam(ctypes.byref(desc), h.data_ptr(), act_in, whead.data_ptr(), wfeat.data_ptr(), n5.weight.data_ptr(),
                                      n5.running_var.data_ptr(), act.data_ptr(), grads.data_ptr(), heat.data_ptr(), maps.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "gradcam")
            assert heat.ndim == 4, 'Batch dimension found in attention map - Must use batch size 1 when computing attention maps'
            att_maps = list(maps.unbind(0))
            self.features, self.grads, self.heat = act, grads, heat
        mm.train(was_training)
        return outputs, att_maps


def add_gradcam(model, output_dir='attention_maps', multimodal=False):
    if multimodal:
        return model.add_gradcam(output_dir)
    raise NotImplementedError("unimodal Grad-CAM in the reference is the third-party `medcam` package (utils/utils.py:455); "
                              "only the multimodal Grad-CAM is part of this path")


def remap_bhb_keys(entries):
    """Key translation of the BHB-10K pretrained DenseNet121 checkpoint (utils/utils.py:368-384): drop the DataParallel prefix and
    address a dense layer's leaves through its `layers` Sequential.  (The result still says `features.*` where this DenseNet says
    `backbone.*` -- SURVEY Appendix A Q13 -- so with strict=False the convolutional weights match nothing, exactly as upstream.)"""
    out = {}
    for key, value in entries.items():
        parts = key.replace('module.', '').split('.')
        if parts[0] == 'features' and len(parts) > 1 and parts[1].startswith('dense'):
            parts.insert(3, 'layers')
        out['.'.join(parts)] = value
    return out


def loadWeights(model, path, device):
    """utils/utils.py:357-390 for local files: plain state_dict, or the BHB-10K pretrained DenseNet121 key remap."""
    checkpoint = torch.load(path, map_location=device)
    if isinstance(checkpoint, dict) and 'model' in checkpoint and path.endswith('DenseNet121_BHB-10K_yAwareContrastive.pth'):
        model.load_state_dict(remap_bhb_keys(checkpoint['model']), strict=False)
        logger.info('Loaded pretrained backbone')
    else:
        model.load_state_dict(checkpoint)
        logger.info('Loaded provided weights from disk')
    return model
