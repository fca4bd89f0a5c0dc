"""ctypes binding of libmmnn_sts.so (include/mmnn_sts.h).  The product path has NO fallback: if the library is
missing or a call fails, an exception is raised."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMNN_LIB_PATH") or os.path.join(_HERE, "libmmnn_sts.so")   # override: developer builds only
_lib = None


class DenseNetConfig(Structure):
    _fields_ = [
        ("in_channels", c_int32), ("init_features", c_int32), ("growth_rate", c_int32), ("bn_size", c_int32),
        ("num_blocks", c_int32), ("block_config", c_int32 * 8), ("eps", c_float), ("momentum", c_float),
        ("dropout_prob", c_float),
    ]


class Conv3dDesc(Structure):
    _fields_ = [("n", c_int32), ("c_in", c_int32), ("d", c_int32), ("h", c_int32), ("w", c_int32), ("c_out", c_int32),
                ("kernel", c_int32 * 3), ("stride", c_int32 * 3), ("padding", c_int32 * 3)]


MLP_MAX_LAYERS = 8
_FP = POINTER(c_float)


class MlpDesc(Structure):
    _fields_ = [
        ("n", c_int32), ("num_layers", c_int32), ("in_dim", c_int32 * MLP_MAX_LAYERS), ("out_dim", c_int32 * MLP_MAX_LAYERS),
        ("relu_first", c_int32 * MLP_MAX_LAYERS), ("dropout_prob", c_float), ("eps", c_float), ("momentum", c_float),
        ("seed", c_uint64), ("training", c_int32), ("first_layer_id", c_int32),
    ]


MULTI_MAX = 64
DT_F64, DT_F32, DT_I64, DT_I32, DT_U8 = 0, 1, 2, 3, 4


class TensorRef(Structure):
    """mmnn_tensor_ref: one small tensor of a multi-tensor launch (mmnn_sgd_step_multi / mmnn_multi_copy)."""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("count", c_int64), ("flat_offset", c_int64), ("first_step", c_int32),
                ("reserved", c_int32)]


class GradcamDesc(Structure):
    _fields_ = [("c_total", c_int32), ("growth", c_int32), ("d", c_int32), ("h", c_int32), ("w", c_int32), ("classes", c_int32),
                ("features", c_int32), ("head_ld", c_int32), ("out_d", c_int32), ("out_h", c_int32), ("out_w", c_int32), ("eps", c_float)]


class MlpParams(Structure):
    _fields_ = [(k, c_void_p * MLP_MAX_LAYERS) for k in (
        "weight", "bias", "gamma", "beta", "running_mean", "running_var", "grad_weight", "grad_bias", "grad_gamma", "grad_beta",
        "num_batches_tracked")]


def lib():
    """Load the shared library once (torch must be imported first so that its HIP runtime is the one bound)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64 with the SONAME the extension needs)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m mmnn_sts_amd.build` (hipcc, gfx950). "
            "mmnn_sts_amd has no CPU / eager fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.mmnn_version.restype = c_int32
    L.mmnn_last_error.restype = c_char_p
    L.mmnn_densenet_plan_create.restype = c_void_p
    L.mmnn_densenet_plan_create.argtypes = [POINTER(DenseNetConfig), c_int32, c_int32, c_int32, c_int32]
    L.mmnn_densenet_plan_destroy.restype = None
    L.mmnn_densenet_plan_destroy.argtypes = [c_void_p]
    for f in ("mmnn_densenet_param_count", "mmnn_densenet_runstat_count", "mmnn_densenet_workspace_bytes"):
        getattr(L, f).restype = c_int64
        getattr(L, f).argtypes = [c_void_p]
    L.mmnn_densenet_out_shape.restype = c_int32
    L.mmnn_densenet_out_shape.argtypes = [c_void_p] + [POINTER(c_int32)] * 4
    L.mmnn_densenet_forward.restype = c_int32
    L.mmnn_densenet_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_uint64, c_void_p]
    L.mmnn_densenet_backward.restype = c_int32
    L.mmnn_densenet_backward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_uint64, c_void_p]
    L.mmnn_densenet_backward_range.restype = c_int32
    L.mmnn_densenet_backward_range.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_uint64, c_int32, c_int32,
                                               c_void_p]
    L.mmnn_densenet_block_param_range.restype = c_int32
    L.mmnn_densenet_block_param_range.argtypes = [c_void_p, c_int32, POINTER(c_int64), POINTER(c_int64)]
    L.mmnn_densenet_relu_mask.restype = c_int32
    L.mmnn_densenet_relu_mask.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    L.mmnn_densenet_ws_offset.restype = c_int64
    L.mmnn_densenet_ws_offset.argtypes = [c_void_p, c_char_p, c_int32, c_int32]
    V, I, U, F = c_void_p, c_int32, c_uint64, c_float
    sigs = {
        "mmnn_gap_linear_forward": [I, I, I, I, V, V, V, V, V, F, U, I, V],
        "mmnn_gap_linear_backward": [I, I, I, I, V, V, V, V, V, V, V, F, U, I, I, V],
        "mmnn_mlp_forward": [POINTER(MlpDesc), POINTER(MlpParams), V, V, V, V],
        "mmnn_mlp_backward": [POINTER(MlpDesc), POINTER(MlpParams), V, V, V, V, V, I, V],
        "mmnn_fusion_heads_forward": [I, I, I, I] + [V] * 9 + [V],
        "mmnn_fusion_heads_backward": [I, I, I, I] + [V] * 14 + [I, V],
        "mmnn_linear_forward": [I, I, I, V, V, V, V, V],
        "mmnn_linear_backward": [I, I, I, V, V, V, V, V, V, I, V],
        "mmnn_cox_blend_loss": [I, I, I, V, V, V, V, V, V, V, V, V],
        "mmnn_cox_blend_loss_typed": [I, I, I, V, V, I, V, I, V, V, V, V, V, V],
        "mmnn_cox_blend_backward": [I, I, I, V, V, V, V, V, V],
        "mmnn_sgd_step": [V, V, V, c_int64, F, F, F, I, I, V],
        "mmnn_sgd_step_multi": [POINTER(TensorRef), I, V, F, F, F, I, V],
        "mmnn_multi_copy": [POINTER(TensorRef), I, V, I, V],
        "mmnn_gradcam": [POINTER(GradcamDesc), V, V, V, V, V, V, V, V, V, V, V],
        "mmnn_bce_logits": [c_int64, I, V, V, V, V, V, V],
        "mmnn_conv3d_out_shape": [POINTER(Conv3dDesc), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)],
        "mmnn_conv3d_forward": [POINTER(Conv3dDesc), V, V, V, V],
        "mmnn_conv3d_backward_data": [POINTER(Conv3dDesc), V, V, V, V],
        "mmnn_conv3d_backward_weight": [POINTER(Conv3dDesc), V, V, V, V, I, V],
        "mmnn_bn3d_forward": [I, I, c_int64, V, V, V, V, V, F, F, I, I, V, F, U, V, V, V, V],
        "mmnn_bn3d_backward": [I, I, c_int64, V, V, V, V, V, I, I, F, U, V, V, V, V, V, V],
        "mmnn_gap_fc_sigmoid_forward": [I, I, c_int64, I, V, V, V, V, V, V],
        "mmnn_gap_fc_sigmoid_backward": [I, I, c_int64, I, V, V, V, V, V, V, V, V],
        "mmnn_densenet_set_timer": [V, I, I],
        "mmnn_densenet_read_timer": [V, POINTER(ctypes.c_double), POINTER(c_int64)],
        "mmnn_densenet_read_timer_class": [V, I, I, POINTER(ctypes.c_double), POINTER(c_int64)],
        "mmnn_densenet_set_option": [V, c_char_p, c_int64],
        "mmnn_densenet_set_batch_counters": [V, V, I],
        "mmnn_measure_mfma_clock": [POINTER(ctypes.c_double), V, V],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.restype = c_int32
        fn.argtypes = args
    L.mmnn_conv3d_wgrad_workspace_bytes.restype = c_int64
    L.mmnn_conv3d_wgrad_workspace_bytes.argtypes = [POINTER(Conv3dDesc)]
    L.mmnn_mlp_saved_floats.restype = c_int64
    L.mmnn_mlp_saved_floats.argtypes = [POINTER(MlpDesc)]
    _lib = L
    return L


def check(status: int, what: str) -> None:
    """C status -> Python exception (1: ValueError, otherwise RuntimeError), SURVEY 8(b) error contract."""
    if status == 0:
        return
    msg = lib().mmnn_last_error().decode("utf-8", "replace")
    if status == 1:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (status {status})")


def last_error() -> str:
    return lib().mmnn_last_error().decode("utf-8", "replace")
