"""Shared helpers for the tests: synthetic state dicts / inputs identical to oracle/make_golden.py."""
import os

import numpy as np
import torch

from oracle import synth
from oracle import restatement as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_CLIN = 32


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def synth_sd(schema, prefix, device="cpu", requires_grad=False):
    sd = {}
    for k, v in synth.synth_state_dict(schema, prefix).items():
        t = torch.from_numpy(np.asarray(v)).to(device)
        if requires_grad and t.is_floating_point() and "running" not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def image_in(n, c, s):
    return torch.from_numpy(synth.uniform(f"image/{n}x{c}x{s}", (n, c, s, s, s)))


def clin_in(n):
    return torch.from_numpy(synth.uniform(f"clinical/{n}", (n, N_CLIN)))


def labels(n):
    if n == 2:
        ev = np.array([[1, 0], [0, 1]], dtype=np.int64)
        du = np.array([[100, 250], [300, 50]], dtype=np.int64)
    else:
        ev = (synth.uniform(f"events/{n}", (n, 2)) > 0).astype(np.int64)
        ev[0, :] = 1
        du = (1 + np.floor((synth.uniform(f"durations/{n}", (n, 2)) * 0.5 + 0.5) * 2998)).astype(np.int64)
    return torch.from_numpy(ev), torch.from_numpy(du)


def stat3(t):
    t = t.detach().double().cpu()
    return np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item()])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
