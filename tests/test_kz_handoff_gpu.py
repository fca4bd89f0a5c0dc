"""Stress of the cross-workgroup K-split hand-off of the small-extent convolutions (csrc/fprop.hpp): partial accumulator
tiles travel between workgroups through HBM behind a ticket.  A missed write-through or a stale read would hand the last arriver an
OLD partial tile -- with identical inputs that is the previous repetition's value and invisible, so every repetition scales the input
differently and is compared with a plan that never splits (plan option "no_kz") on the same input.  The extents make the dispatcher pick
kz = 2, 4 and 8 (fprop_dispatch.hpp: tiles * mtiles * kz <= 256)."""
import ctypes

import pytest
import torch

from oracle import restatement as R
from oracle import synth
from tests._util import synth_sd

pytestmark = pytest.mark.gpu


def _pair(cfg, n, dhw):
    from mmnn_sts_amd import _lib
    from tests._native import NativeBackbone
    a, b = NativeBackbone(cfg, n, *dhw), NativeBackbone(cfg, n, *dhw)
    _lib.check(b.L.mmnn_densenet_set_option(b.plan, b"no_kz", 1), "set_option")
    return a, b


@pytest.mark.parametrize("blocks,dhw,n", [((2, 3, 3), (32, 32, 32), 2), ((2, 2, 4, 3), (64, 64, 64), 1), ((3, 3), (24, 20, 36), 3)])
def test_kz_handoff_matches_unsplit(blocks, dhw, n):
    cfg = R.DenseNetCfg(in_channels=2, block_config=blocks)
    sch = R.densenet_schema(cfg)
    kz, ref = _pair(cfg, n, dhw)
    flat, run = kz.flatten(synth_sd(sch, "densenet."))
    run2 = run.clone()
    x0 = torch.from_numpy(synth.uniform(f"kz/{blocks}", (n, 2) + dhw)).cuda()
    cot = torch.from_numpy(synth.uniform("kz/cot", kz.out_shape)).cuda()
    worst = 0.0
    for rep in range(12):
        x = (x0 * (1.0 + 0.03 * rep)).contiguous()
        out = kz.forward(flat, run, x, training=True)
        g = kz.backward(flat, x, cot)
        out_r = ref.forward(flat, run2, x, training=True)
        g_r = ref.backward(flat, x, cot)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and torch.isfinite(g).all()
        # same arithmetic up to the association of the K sum (partial sums per slice instead of one chain): fp32 round-off only
        eo = float((out - out_r).abs().max() / out_r.abs().max())
        eg = float((g - g_r).norm() / g_r.norm())
        worst = max(worst, eo, eg)
        assert eo < 2e-4 and eg < 2e-4, (rep, eo, eg)
        # and reproducible: the slices are summed in slice order whatever the arrival order
        out2 = kz.forward(flat, run.clone(), x, training=True)
        g2 = kz.backward(flat, x, cot)
        assert torch.equal(out, out2) and torch.equal(g, g2), rep
    print("worst relative deviation from the unsplit plan", worst)
