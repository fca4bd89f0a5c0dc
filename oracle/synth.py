"""Counter-based synthetic tensors shared by the oracle, the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Weights and inputs of every golden fixture are
produced by this formula from (tensor name, element index), so fixtures store only expected OUTPUTS;
the same formula is evaluated on the GPU box without needing torch's CPU RNG streams.

The value of element i of a tensor named `name` is   scale * (2*u - 1) + offset   with
u = splitmix64(i * GOLDEN + crc32(name) * MIX) / 2**64  (53 high bits), i.e. uniform in [-scale, scale).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def uniform(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """float32 tensor of `shape`, uniform in [offset-scale, offset+scale), a pure function of (name, index)."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(name_seed(name)) * np.uint64(0xD1B54A32D192ED03)) & _M64
    h = _splitmix64(key)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return ((2.0 * u - 1.0) * scale + offset).astype(np.float32).reshape(shape)


def synth_state_dict(shapes: dict, prefix: str = "") -> dict:
    """Deterministic, non-trivial values for every entry of a {key: shape} schema.

    conv / linear weights: variance of kaiming-normal (2/fan_in) resp. 1/fan_in, uniform shape;
    BN weight in [0.7,1.3), BN bias in [-0.2,0.2), running_mean in [-0.1,0.1), running_var in [0.8,1.2);
    linear bias in [-0.1,0.1); num_batches_tracked = 0.
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        full = prefix + key
        leaf = key.split(".")[-1]
        parent = key.split(".")[-2] if "." in key else ""
        is_bn = parent.startswith("norm") or parent.startswith("bn") or (leaf in ("weight", "bias") and len(shape) == 1 and (
            key.rsplit(".", 1)[0] + ".running_mean") in shapes)
        if leaf == "num_batches_tracked":
            out[key] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = uniform(full, shape, 0.1)
        elif leaf == "running_var":
            out[key] = uniform(full, shape, 0.2, 1.0)
        elif is_bn and leaf == "weight":
            out[key] = uniform(full, shape, 0.3, 1.0)
        elif is_bn and leaf == "bias":
            out[key] = uniform(full, shape, 0.2)
        elif leaf == "weight" and len(shape) == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            out[key] = uniform(full, shape, float(np.sqrt(6.0 / fan_in)))
        elif leaf == "weight" and len(shape) == 2:
            out[key] = uniform(full, shape, float(np.sqrt(3.0 / shape[1])))
        elif leaf == "bias":
            out[key] = uniform(full, shape, 0.1)
        else:
            raise KeyError(f"no synthesis rule for {key} {shape}")
    return out
