"""MI355X-native mirror of the late-fusion model (models/multimodal.py:9-90 of DigITs-AIML/MMNN_STS)."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.utils import BackpropagatableFeatureExtractor, MultiModalGradCAM
from .mlp import MLP


class MultiModalModel(nn.Module):
    """image encoder + clinical MLP, 12+12 features -> fused head; with `blend`, per-modality heads and a (3, N, C) stack
    [fused, image, clinical] (models/multimodal.py:51-80).  The heads/concat/stack run as one fused HIP kernel."""

    def __init__(self, image_model, clinical_predictors, num_classes, num_features, blend=False):
        super().__init__()
        self.image_model = image_model
        self.clinical_predictors = clinical_predictors
        self.num_classes = num_classes
        self.num_features = num_features
        self.num_clinical_inputs = len(clinical_predictors)
        self.clinical_model = MLP(self.num_clinical_inputs, self.num_classes, self.num_features)
        self.output_head = nn.Linear(self.num_features * 2, self.num_classes)
        self.blend = blend
        self.image_model = BackpropagatableFeatureExtractor(self.image_model)
        self.clinical_model = BackpropagatableFeatureExtractor(self.clinical_model)
        self.clinical_output_head = nn.Linear(self.num_features, self.num_classes)
        self.image_output_head = nn.Linear(self.num_features, self.num_classes)

    def forward(self, x):
        image_features = self.image_model(x["image"])
        clinical_features = self.clinical_model(x["clinical"])
        return ops.FusionHeads.apply(image_features, clinical_features, self.output_head.weight, self.output_head.bias,
                                     self.image_output_head.weight, self.image_output_head.bias,
                                     self.clinical_output_head.weight, self.clinical_output_head.bias, bool(self.blend))

    @property
    def gradcam_layer(self):
        return self.image_model.model.backbone

    def add_gradcam(self, output_dir):
        return MultiModalGradCAM(self)
