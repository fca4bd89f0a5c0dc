"""White-box parity of the HIP DenseNet backbone (forward intermediates, running statistics, every parameter
gradient) against the CPU oracle on the same synthetic inputs.  fp32; tolerances written per check."""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import synth
from tests._util import rel_err, synth_sd

pytestmark = pytest.mark.gpu

CASES = [
    # (in_ch, block_config, (D,H,W), N)
    (2, (6, 12, 24, 16), (64, 64, 64), 2),
    (1, (6, 12, 24, 16), (64, 64, 64), 2),
    (2, (2, 2, 2), (40, 36, 44), 3),        # ragged, non power-of-two extents; odd remainders in every pool
    (2, (6, 12, 4), (64, 64, 64), 1),       # TinyDensenet layout, single sample
    (3, (2, 2), (40, 44, 48), 2),           # stem kernels are instantiated per input-channel count (csrc/stem.hip): odd count, no channel pairing
    (4, (2, 2), (36, 40, 44), 2),           # four channels: two LDS buffers of the conv0 forward at the 160 KB limit
]


def _run_case(in_ch, blocks, dhw, n, full=True):
    """Truth = the oracle evaluated in fp64 (the fp32 oracle itself is only within ~3e-5 of it at 64^3, see
    DESIGN.md "Tolerances"); bar = 1e-4 relative (north star) on every intermediate, the output and the running
    statistics; gradients: per-tensor L2 error <= 1e-3*|g| + 1e-5*|g_all| (several gradients are analytically 0).
    `full=False`: extents whose last block has a single voxel per sample -- two-sample batch norm is ill-conditioned
    there (fp32 and fp64 oracles disagree by >10 %), so only the blocks before it are compared."""
    from tests._native import NativeBackbone, backbone_run_keys
    cfg = R.DenseNetCfg(in_channels=in_ch, block_config=blocks)
    sch = R.densenet_schema(cfg)
    x = torch.from_numpy(synth.uniform(f"bb/{n}x{in_ch}x{dhw}", (n, in_ch) + dhw))
    nb = NativeBackbone(cfg, n, *dhw)
    flat, run = nb.flatten(synth_sd(sch, "densenet."))
    xg = x.cuda()
    out = nb.forward(flat, run, xg, training=True)
    torch.cuda.synchronize()
    # fp64 oracle taking the SAME ReLU branches as the device (ReLU'(0) is a convention; near-zero pre-activations would
    # otherwise make any two fp32 implementations disagree by percents in block 4, where a channel has N*V = 16 samples)
    masks = nb.relu_masks(flat) if full else None
    sd = {k: (v.double().requires_grad_("running" not in k) if v.is_floating_point() else v)
          for k, v in synth_sd(sch, "densenet.").items()}
    taps = {}
    h = R.densenet_backbone(sd, x.double(), cfg, True, taps=taps, relu_masks=masks)
    cot = torch.from_numpy(synth.uniform("bb/cot", tuple(h.shape)))
    if full:
        (h * cot.double()).sum().backward()
    errs = {}
    c0 = taps["conv0"]
    errs["conv0"] = rel_err(nb.region("conv0", tuple(c0.shape)).cpu().numpy(), c0.detach().numpy())
    nblk = len(blocks) if full else len(blocks) - 1
    for b in range(nblk):
        ref = taps[f"block{b + 1}"].detach()
        errs[f"block{b + 1}"] = rel_err(nb.region("x", tuple(ref.shape), b).cpu().numpy(), ref.numpy())
    if full:
        errs["norm5"] = rel_err(out.cpu().numpy(), h.detach().numpy())
    assert torch.isfinite(out).all()
    for k, e in errs.items():
        assert e < 1e-4, (k, errs)
    if not full:
        return errs, None
    got_run = nb.unflatten(run.cpu(), backbone_run_keys(sch))
    for k, v in got_run.items():
        assert rel_err(v.numpy(), sd[k].detach().numpy()) < 1e-4, k
    g = nb.backward(flat, xg, cot.cuda())
    torch.cuda.synchronize()
    got = nb.unflatten(g.cpu())
    gl2 = float(torch.sqrt(sum((sd[k].grad ** 2).sum() for k in got)))
    worst = (0.0, "")
    bad = []
    for k, v in got.items():
        ref = sd[k].grad
        err = float((v.double() - ref).norm())
        tol = 1e-3 * float(ref.norm()) + 1e-5 * gl2
        worst = max(worst, (err / tol, k))
        if err > tol:
            bad.append((k, err, float(ref.norm()), float(v.double().norm())))
    assert not bad, (len(bad), len(got), gl2, bad[:3], bad[-12:])
    return errs, worst


@pytest.mark.parametrize("in_ch,blocks,dhw,n", CASES)
def test_backbone_forward_backward(in_ch, blocks, dhw, n):
    errs, worst = _run_case(in_ch, blocks, dhw, n)
    print("forward rel errors", errs, "worst gradient (err/tol, name)", worst)


# Tile-instantiation matrix.  Tile shapes are picked from the extent (csrc/fprop_dispatch.hpp: `dispatch`, csrc/wgrad.hip:
# `wg3_tile`, `wg1_wc`), so every branch needs an extent of its own.  Block-1 extents of the cases (block 2 = half of it):
#   20^3 (N=4)    W > 16: wave-specialised conv2 forward <27,..,2,4,32,true>, KC=2 conv2 data-grad <27,PRO_GRAD,..,1,4,32>,
#                 wgrad3<1,2,32>; 252 voxel tiles -> 128-wide 1x1x1 tile for conv1 forward / data-grad; block 2: W = 10 tiles
#   6x10x17       ragged W just above the 16 boundary, odd rows (no 16-byte staging: scalar path of the W > 16 tiles)
#   4x6x33        W = 33: one full 32-wide tile + a 1-voxel remainder tile in every row
#   12x12x40      64-wide 1x1x1 tile (blocks_a < 192 <= blocks_b) incl. the PRO_NONE transition conv at 20x6x6
# The BASELINE extents themselves (32^3 .. 4^3 at N = 2) run in test_baseline_config3_backbone_128.
MATRIX = [
    (2, (2, 2), (80, 80, 80), 4),
    (2, (2, 2), (24, 40, 66), 2),
    (1, (2, 2), (16, 24, 130), 2),
    (2, (3, 2), (48, 48, 160), 2),
]


@pytest.mark.parametrize("in_ch,blocks,dhw,n", MATRIX)
def test_backbone_tile_matrix(in_ch, blocks, dhw, n):
    errs, worst = _run_case(in_ch, blocks, dhw, n)
    print("forward rel errors", errs, "worst gradient (err/tol, name)", worst)


def test_baseline_config3_backbone_128():
    """The backbone of BASELINE configs[2] at its own size (2 x 2 x 128^3): every intermediate, the running statistics and all
    364 backbone gradients against the fp64 oracle -- this is the extent whose W > 16 tiles the benchmark dispatches to."""
    errs, worst = _run_case(2, (6, 12, 24, 16), (128, 128, 128), 2)
    print("forward rel errors", errs, "worst gradient (err/tol, name)", worst)


def test_baseline_config2_backbone_128():
    """BASELINE configs[1]: single-channel (t1) volumes, 2 x 1 x 128^3."""
    errs, worst = _run_case(1, (6, 12, 24, 16), (128, 128, 128), 2)
    print("forward rel errors", errs, "worst gradient (err/tol, name)", worst)


def test_backbone_minimum_extent():
    """32^3 is the smallest legal input: every block must run (1x1x1 voxels in block 4)."""
    _run_case(2, (6, 12, 24, 16), (32, 32, 32), 2, full=False)


def test_backbone_accumulate_and_eval():
    from tests._native import NativeBackbone
    cfg = R.DenseNetCfg(in_channels=2, block_config=(2, 2))
    sch = R.densenet_schema(cfg)
    sd = synth_sd(sch, "densenet.")
    x = torch.from_numpy(synth.uniform("bb/acc", (2, 2, 24, 24, 24)))
    with torch.no_grad():
        ref = R.densenet_backbone(sd, x, cfg, False)
    nb = NativeBackbone(cfg, 2, 24, 24, 24)
    flat, run = nb.flatten(synth_sd(sch, "densenet."))
    run0 = run.clone()
    out = nb.forward(flat, run, x.cuda(), training=False)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-5
    assert torch.equal(run, run0)                      # eval never touches the running statistics
    out = nb.forward(flat, run, x.cuda(), training=True)
    cot = torch.ones_like(out)
    g1 = nb.backward(flat, x.cuda(), cot).clone()
    g2 = nb.backward(flat, x.cuda(), cot, accumulate=True, grad=g1.clone())
    torch.cuda.synchronize()
    assert rel_err(g2.cpu().numpy(), (2 * g1).cpu().numpy()) < 1e-6
    # run-to-run: statistics are fp64-accumulated, weight gradients are slab sums (no fp32 atomics on tensors)
    out2 = nb.forward(flat, run, x.cuda(), training=True)
    g3 = nb.backward(flat, x.cuda(), cot)
    torch.cuda.synchronize()
    assert rel_err(g3.cpu().numpy(), g1.cpu().numpy()) < 1e-5 and rel_err(out2.cpu().numpy(), out.cpu().numpy()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("in_ch", [1, 2])
def test_backbone_repeated_calls_are_bit_identical(in_ch):
    """One plan, many calls: every forward / backward of the same inputs must reproduce the first bit for bit, whatever ran
    before it (the kernels keep no state between calls and never read LDS or workspace words they did not write)."""
    from tests._native import NativeBackbone
    cfg = R.DenseNetCfg(in_channels=in_ch)
    n, s = 2, 64
    nb = NativeBackbone(cfg, n, s, s, s)
    flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(n, in_ch, s, s, s, device="cuda", generator=g)
    cot = torch.randn(nb.out_shape, device="cuda", generator=g)
    ref_o = ref_g = None
    for op in "FFBFBBFFB":
        if op == "F":
            o = nb.forward(flat, run.clone(), x, True, seed=1)
            assert torch.isfinite(o).all()
            if ref_o is None:
                ref_o = o
            assert torch.equal(o, ref_o), f"forward deviates by {float((o - ref_o).abs().max())}"
        else:
            gr = nb.backward(flat, x, cot, seed=1)
            assert torch.isfinite(gr).all()
            if ref_g is None:
                ref_g = gr
            assert torch.equal(gr, ref_g), f"backward deviates by {float((gr - ref_g).abs().max())}"


@pytest.mark.gpu
@pytest.mark.parametrize("side", [1, 2])
def test_backbone_side_streams_are_equivalent(side):
    """Plan option "side_streams": the weight-gradient kernels on 1 or 2 side streams (event hand-offs) give bit-identical gradients
    to the default single-stream schedule."""
    import ctypes
    from mmnn_sts_amd import _lib
    from tests._native import NativeBackbone
    cfg = R.DenseNetCfg(in_channels=2)
    n, s = 2, 64
    nb = NativeBackbone(cfg, n, s, s, s, dropout=0.2)
    flat, run = nb.flatten(synth_sd(R.densenet_schema(cfg), "densenet."))
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, 2, s, s, s, device="cuda", generator=g)
    cot = torch.randn(nb.out_shape, device="cuda", generator=g)
    nb.forward(flat, run.clone(), x, True, seed=9)
    ref = nb.backward(flat, x, cot, seed=9).clone()
    _lib.check(_lib.lib().mmnn_densenet_set_option(nb.plan, b"side_streams", side), "set_option")
    for _ in range(2):
        nb.forward(flat, run.clone(), x, True, seed=9)
        got = nb.backward(flat, x, cot, seed=9)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all() and torch.equal(got, ref), float((got - ref).abs().max())
    _lib.check(_lib.lib().mmnn_densenet_set_option(nb.plan, b"side_streams", 0), "set_option")
    nb.forward(flat, run.clone(), x, True, seed=9)
    assert torch.equal(nb.backward(flat, x, cot, seed=9), ref)


# Persistent block forward (csrc/blockfwd.hip; opt-in experiment, plan option "persistent_forward"): the 8^3 / 4^3 dense blocks run as ONE
# resident launch with grid barriers between the layers' phases.  Cases: (blocks, input extent, N) -> extents of the persistent blocks
#   (2,2,3,2) 128^3 N=1   block 3: 8^3 (64 workgroups, chip-wide barrier), block 4: 4^3 (8 workgroups on one XCD)
#   (2,2,3,2) 128^3 N=3   192 / 24 workgroups (tiles of three samples)
#   (2,2,2) 64x128x256 N=2  block 3: 4x8x16 (W = 16: one-row tiles), block 2 (W = 32) stays on the per-layer kernels
PERSISTENT = [((2, 2, 3, 2), (128, 128, 128), 1), ((2, 2, 3, 2), (128, 128, 128), 3), ((2, 2, 2), (64, 128, 256), 2)]


@pytest.mark.parametrize("blocks,dhw,n", PERSISTENT)
def test_persistent_block_forward_matches_layer_kernels(blocks, dhw, n):
    """Same arithmetic, other summation order (K split over the waves of one workgroup instead of over workgroups): every tensor the
    block produces -- bottleneck tensors, concat buffer, output, running statistics -- agrees with the per-layer kernels to fp32
    round-off, in training and in eval mode; the gradients of a backward that consumes the persistent forward's tensors agree up to ReLU
    branch flips of near-zero pre-activations; a repeated launch reproduces itself bit for bit; the barrier error flag stays clear."""
    from mmnn_sts_amd import _lib
    from tests._native import NativeBackbone
    cfg = R.DenseNetCfg(in_channels=2, block_config=blocks)
    sch = R.densenet_schema(cfg)
    a, b = NativeBackbone(cfg, n, *dhw), NativeBackbone(cfg, n, *dhw)
    _lib.check(a.L.mmnn_densenet_set_option(a.plan, b"persistent_forward", 1), "set_option")      # opt-in experiment (default: per-layer kernels)
    flat, run_a = a.flatten(synth_sd(sch, "densenet."))
    run_b = run_a.clone()
    x = torch.from_numpy(synth.uniform(f"pb/{blocks}/{n}", (n, 2) + dhw)).cuda()
    cot = torch.from_numpy(synth.uniform("pb/cot", a.out_shape)).cuda()
    for training in (True, False):
        oa = a.forward(flat, run_a, x, training=training, seed=11)
        ob = b.forward(flat, run_b, x, training=training, seed=11)
        torch.cuda.synchronize()
        assert torch.isfinite(oa).all()
        assert rel_err(oa.cpu().numpy(), ob.cpu().numpy()) < 2e-5, training
        dims = [(s - 1) // 2 + 1 for s in dhw]
        dims = [(s - 1) // 2 + 1 for s in dims]
        c = cfg.init_features
        for blk, nl in enumerate(blocks):
            ctot = c + nl * cfg.growth_rate
            xa = a.region("x", (n, ctot, *dims), blk)
            xb = b.region("x", (n, ctot, *dims), blk)
            assert rel_err(xa.cpu().numpy(), xb.cpu().numpy()) < 2e-5, (training, blk)
            for l in range(nl):
                ta = a.region("t1", (n, cfg.bn_size * cfg.growth_rate, *dims), blk, l)
                tb = b.region("t1", (n, cfg.bn_size * cfg.growth_rate, *dims), blk, l)
                assert rel_err(ta.cpu().numpy(), tb.cpu().numpy()) < 2e-5, (training, blk, l)
            c, dims = ctot // 2, [s // 2 for s in dims]
        if training:
            assert rel_err(run_a.cpu().numpy(), run_b.cpu().numpy()) < 1e-5
            ga = a.backward(flat, x, cot, seed=11)
            gb = b.backward(flat, x, cot, seed=11)
            oa2 = a.forward(flat, run_a.clone(), x, training=True, seed=11)
            ga2 = a.backward(flat, x, cot, seed=11)
            torch.cuda.synchronize()
            assert float((ga - gb).norm() / gb.norm()) < 3e-2
            assert torch.equal(oa, oa2) and torch.equal(ga, ga2)
    off = a.L.mmnn_densenet_ws_offset(a.plan, b"blk_sync", 0, 0)
    assert off >= 0
    words = a.ws[off:off + 64].view(torch.int32).cpu()
    assert int(words[1::2].abs().sum()) == 0, words            # no barrier gave up
    assert int(words[0::2].sum()) > 0                           # ... and the persistent kernels really ran


def test_persistent_block_forward_dropout_streams_match():
    """Channel dropout inside the persistent launch uses the same counter-based stream (seed, layer, sample, channel) as the per-layer
    kernels: with p = 0.3 both plans must drop exactly the same channels."""
    from mmnn_sts_amd import _lib
    from tests._native import NativeBackbone
    cfg = R.DenseNetCfg(in_channels=2, block_config=(2, 2, 3, 2))
    sch = R.densenet_schema(cfg)
    a, b = NativeBackbone(cfg, 2, 128, 128, 128, dropout=0.3), NativeBackbone(cfg, 2, 128, 128, 128, dropout=0.3)
    _lib.check(a.L.mmnn_densenet_set_option(a.plan, b"persistent_forward", 1), "set_option")
    flat, run = a.flatten(synth_sd(sch, "densenet."))
    x = torch.from_numpy(synth.uniform("pb/drop", (2, 2, 128, 128, 128))).cuda()
    oa = a.forward(flat, run.clone(), x, training=True, seed=77)
    ob = b.forward(flat, run.clone(), x, training=True, seed=77)
    torch.cuda.synchronize()
    c = cfg.init_features
    for nl in cfg.block_config[:-1]:
        c = (c + nl * cfg.growth_rate) // 2
    ctot = c + cfg.block_config[-1] * cfg.growth_rate          # channels of the last block's concat buffer (4^3 voxels at 128^3)
    xa, xb = a.region("x", (2, ctot, 4, 4, 4), 3), b.region("x", (2, ctot, 4, 4, 4), 3)
    za, zb = (xa.abs().sum(dim=(2, 3, 4)) == 0), (xb.abs().sum(dim=(2, 3, 4)) == 0)
    assert torch.equal(za, zb) and 0 < int(za.sum()) < za.numel() // 2
    assert rel_err(oa.cpu().numpy(), ob.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("mode", ["0", "16"])
def test_bf16x3_switch_positions_keep_parity(mode):
    """The suite runs with the default kernel selection (conv2 forward / data gradient of extents wider than 16 voxels on three-piece bf16
    MFMAs, csrc/conv3_bf16x3.hip).  The switches are read once per process, so the other positions run in a fresh one: MMNN_BF16X3=0 (the
    fp32-MFMA kernels for every extent) and MMNN_BF16X3=16 (the opt-in 16-voxel tile as well) must pass the same tile-matrix parity
    (ragged W = 17 / 33 included) against the fp64 oracle at the same tolerances."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MMNN_BF16X3=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "tile_matrix"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
