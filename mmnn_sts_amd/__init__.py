"""mmnn_sts_amd -- MI355X (gfx950) native implementation of the MMNN_STS multimodal-fusion training path.

Python mirror of the reference's module interface (`models.densenet`, `models.mlp`, `models.multimodal`,
`losses.GradientBlender`, `losses.losses`, `utils.utils`) over a C-ABI library of hand-written HIP kernels
(include/mmnn_sts.h, built by `python -m mmnn_sts_amd.build`).  There is no CPU or eager fallback.
"""
__version__ = "0.1.0"
