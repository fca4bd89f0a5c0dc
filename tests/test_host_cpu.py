"""`not gpu` tests of the host side: the C-ABI library loads and exports every symbol include/mmnn_sts.h declares, plan
queries (pure host arithmetic), the Python mirror's module tree / state_dict schema, the blender's generic (non-native)
path, the refusal to run without a GPU, and the N > 1 gradient exchange on gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests._util import N_CLIN, labels, load_golden, synth_sd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mmnn_sts_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "mmnn_sts.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mmnn_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/mmnn_sts.h but not exported"
    assert lib.mmnn_version() >= 100


def test_plan_queries_without_gpu(lib):
    from mmnn_sts_amd import _lib
    cfg = _lib.DenseNetConfig(2, 64, 32, 4, 4, (ctypes.c_int32 * 8)(6, 12, 24, 16, 0, 0, 0, 0), 1e-5, 0.1, 0.2)
    p = lib.mmnn_densenet_plan_create(ctypes.byref(cfg), 2, 128, 128, 128)
    assert p
    assert lib.mmnn_densenet_param_count(p) == 11_276_902 - (1024 * 12 + 12) - (12 * 2 + 2)
    assert lib.mmnn_densenet_runstat_count(p) == 83_769 - 121 - 0      # every BN buffer except the 121 batch counters
    c = [ctypes.c_int32() for _ in range(4)]
    assert lib.mmnn_densenet_out_shape(p, *[ctypes.byref(v) for v in c]) == 0
    assert [v.value for v in c] == [1024, 4, 4, 4]
    assert lib.mmnn_densenet_workspace_bytes(p) > 0
    assert lib.mmnn_densenet_ws_offset(p, b"x", 0, 0) >= 0 and lib.mmnn_densenet_ws_offset(p, b"nope", 0, 0) == -1
    lib.mmnn_densenet_plan_destroy(p)
    # error contract: status / NULL + thread-local message, never abort
    bad = _lib.DenseNetConfig(9, 64, 32, 4, 4, (ctypes.c_int32 * 8)(6, 12, 24, 16, 0, 0, 0, 0), 1e-5, 0.1, 0.2)
    assert not lib.mmnn_densenet_plan_create(ctypes.byref(bad), 2, 128, 128, 128)
    assert b"in_channels" in lib.mmnn_last_error()
    small = lib.mmnn_densenet_plan_create(ctypes.byref(cfg), 2, 8, 8, 8)
    assert not small and b"too small" in lib.mmnn_last_error()
    assert lib.mmnn_densenet_forward(None, None, None, None, None, None, 1, 0, None) == 1


def test_module_tree_matches_reference_schema():
    from mmnn_sts_amd.models.densenet import DenseNet121, TinyDensenet
    from mmnn_sts_amd.models.mlp import MLP
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    img = DenseNet121(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=0.2)
    mm = MultiModalModel(img, [f"p{i}" for i in range(N_CLIN)], 2, 12, blend=True)
    sch = R.multimodal_schema(R.DenseNetCfg(), N_CLIN, 2, 12)
    sd = mm.state_dict()
    assert list(sd.keys()) == list(sch.keys()) and len(sd) == 779
    assert all(tuple(sd[k].shape) == tuple(sch[k]) for k in sch)
    assert sum(p.numel() for p in mm.parameters()) == 11_279_170 and len(list(mm.parameters())) == 398
    assert hasattr(img, "backbone") and hasattr(img, "features") and hasattr(img, "class_layers")
    assert mm.gradcam_layer is img.backbone
    # load_state_dict keeps the flat storage; reference init: BN gamma=1 beta=0, Linear bias 0
    bb = img.backbone
    bb._flatten()
    mm.load_state_dict(synth_sd(sch, "fusion."))
    assert bb._storage_ok()
    assert float(bb.conv0.weight.flatten()[0]) == float(bb._flat[0])
    fresh = TinyDensenet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12)
    assert float(fresh.backbone.norm0.weight.min()) == 1.0 and float(fresh.features.feature_layer.bias.abs().max()) == 0.0
    assert list(MLP(32, 2, 12).state_dict().keys()) == list(R.mlp_schema(32, 2, 12).keys())
    with pytest.raises(NotImplementedError):
        DenseNet121(spatial_dims=2, in_channels=2, out_channels=2, feature_channels=12)


def test_no_cpu_fallback():
    from mmnn_sts_amd.models.densenet import TinyDensenet
    from mmnn_sts_amd.models.mlp import MLP
    m = TinyDensenet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 1, 32, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        MLP(32, 2, 12)(torch.zeros(2, 32))
    import mmnn_sts_amd
    src = "".join(open(os.path.join(os.path.dirname(mmnn_sts_amd.__file__), f)).read()
                  for f in ("ops.py", "_lib.py", "optim.py", "distributed.py", "models/densenet.py", "models/mlp.py",
                            "models/multimodal.py", "losses/GradientBlender.py", "losses/losses.py", "utils/utils.py"))
    assert "import oracle" not in src and "from oracle" not in src          # the product never touches the checker


def test_blender_generic_path_and_errors():
    """A non-native loss callable goes through surv_criterion head by head, like the reference (CPU tensors allowed)."""
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    from mmnn_sts_amd.utils.utils import surv_criterion
    g = load_golden("g6_blender.npz")
    preds = torch.tensor([[[.3, -.2], [.1, .4]], [[.5, 0], [-.1, .2]], [[0, .1], [.2, -.3]]], requires_grad=True)
    ev, du = labels(2)
    gb = GradientBlender(R.CoxPH, survival=True, surv_criterion=surv_criterion)
    loss, sel = gb.computeLoss(preds, ev, du)
    assert abs(loss.item() - g["kat3_loss"][0]) < 1e-6 and abs(sel.item() - g["kat3_sel"][0]) < 1e-6
    loss.backward()
    assert preds.grad is not None and len(gb.history) == 0
    gb.updateWeights(preds.detach(), ev, du, preds.detach(), ev, du)
    assert len(gb.history) == 1 and abs(float(gb.weights.sum()) - 1.0) < 1e-6
    with pytest.raises(ValueError):
        GradientBlender(R.CoxPH, survival=True, reduction="bogus", surv_criterion=surv_criterion).computeLoss(preds, ev, du)
    assert abs(float(GradientBlender(R.CoxPH, survival=True, reduction="mean", surv_criterion=surv_criterion)
                     .computeLoss(preds.detach(), ev, du)[0]) - g["kat3_loss"][0] / 3) < 1e-6


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MMNN_ROOT"])
from mmnn_sts_amd import distributed as D
rank, world, _ = D.init_from_env("gloo")
torch.manual_seed(0)
class FakeBackbone(torch.nn.Module):
    """stands in for the native backbone: parameters are views of a flat buffer, .flat_grad is one bucket"""
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(70000)); self.b = torch.nn.Parameter(torch.zeros(5))
        self._g = torch.zeros(70005)
    @property
    def flat_parameters(self): return torch.cat([self.w.data, self.b.data])
    @property
    def flat_grad(self): return self._g
net = torch.nn.Sequential(FakeBackbone(), torch.nn.Linear(4, 3))
D.broadcast_parameters(net)
# every rank owns different micro-batches: its local gradient sums
micro = [torch.full((70005,), float(10 * rank + k + 1)) for k in range(2)]
net[0]._g.copy_(sum(micro))
net[1].weight.grad = torch.full((3, 4), float(rank + 1)); net[1].bias.grad = torch.full((3,), float(rank + 1) * 2)
D.allreduce_gradients(net)
expect = sum(float(10 * r + k + 1) for r in range(world) for k in range(2))
assert torch.allclose(net[0]._g, torch.full((70005,), expect)), (rank, net[0]._g[:3])
assert torch.allclose(net[1].weight.grad, torch.full((3, 4), float(sum(r + 1 for r in range(world)))))
assert torch.allclose(net[1].bias.grad, torch.full((3,), 2.0 * sum(r + 1 for r in range(world))))
# the overlapped reducer: ranges reported from "inside the backward" are reduced at once, the rest in finish(); un-armed windows
# reduce everything in finish().  Both must give exactly what allreduce_gradients gives.
class HookedBackbone(FakeBackbone):
    def set_grad_ready_hook(self, hook): self._hook = hook
    def fake_backward(self, val):
        self._g.fill_(val)
        for b, e in ((40000, 70005), (10000, 40000), (0, 10000)):      # last block first, like the native backbone
            if self._hook is not None: self._hook(self, b, e)
net2 = torch.nn.Sequential(HookedBackbone(), torch.nn.Linear(4, 3))
red = D.OverlappedGradientReducer(net2)
for armed in (True, False, True):
    if armed: red.arm()
    net2[0].fake_backward(float(rank + 1))
    net2[1].weight.grad = torch.full((3, 4), float(rank + 1)); net2[1].bias.grad = None
    assert len(red._works) == (3 if armed else 0)
    red.finish()
    tot = float(sum(r + 1 for r in range(world)))
    assert torch.equal(net2[0]._g, torch.full((70005,), tot)), (armed, net2[0]._g[:3])
    assert torch.equal(net2[1].weight.grad, torch.full((3, 4), tot))
    assert not red._works and not red._done and not red._armed
red.detach()
assert net2[0]._hook is None
# max-over-ranks timing reduction used by bench.py
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == world
dist.destroy_process_group()
print("ok", rank)
'''


def test_gradient_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, MMNN_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_kz_handoff_isa():
    """the fence-free cross-workgroup K-split hand-off (csrc/fprop.hpp, MMNN_KZ_FENCED == 0) is only valid if the partial
    tiles really leave with write-through stores and are read back with L1-bypassing loads.  Disassemble the gfx950 code objects of the
    convolution translation units and check, for every kernel that takes a ticket (a returning agent-scope global_atomic_add), that
    `global_store_dword ... sc1` precedes it and `global_load_dword ... sc1` follows it (every wave group publishes its 16 / KS rows of
    the tile -- 16 where one group holds the whole tile -- and reads them back from up to 8 slices: at least 2 stores and 16 loads in
    the code of a kernel)."""
    import re
    import shutil
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("ROCm LLVM tools not present")
    objs = sorted(f for f in os.listdir(os.path.join(ROOT, "mmnn_sts_amd", "build")) if f.startswith("fprop_inst_") and f.endswith(".o"))
    assert len(objs) >= 6, "build the library first (python -m mmnn_sts_amd.build)"
    tmp = tempfile.mkdtemp()
    checked = 0
    try:
        for o in objs:
            shutil.copy(os.path.join(ROOT, "mmnn_sts_amd", "build", o), os.path.join(tmp, o))
            r = subprocess.run([objdump, "--offloading", os.path.join(tmp, o)], capture_output=True, text=True)     # extracts the bundles next to the file
            cos = [f for f in os.listdir(tmp) if f.startswith(o + ".") and "gfx950" in f]
            assert r.returncode == 0 and len(cos) == 1, (r.stderr, os.listdir(tmp))
            co = os.path.join(tmp, cos[0])
            asm = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
            for name, body in re.findall(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, flags=re.S | re.M):
                if "fprop_kernel" not in name:
                    continue
                lines = body.splitlines()
                tickets = [i for i, l in enumerate(lines) if "global_atomic_add " in l and "sc0" in l]     # returning 32-bit add = the ticket
                if not tickets:
                    continue
                t = tickets[0]
                st = sum(1 for l in lines[:t] if "global_store_dword " in l and " sc1" in l)
                ld = sum(1 for l in lines[t:] if "global_load_dword " in l and " sc1" in l)
                assert st >= 2 and ld >= 16, (o, name[:80], st, ld)
                checked += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert checked >= 8, checked          # every small-extent tile (KZ_OK in fprop_dispatch.hpp) of the six instantiations


def test_constructor_limits_are_reported_at_construction():
    """the native kernels serve a narrower constructor contract than the reference's DenseNet;
    a configuration outside it must fail when the module is BUILT, with a ValueError, not at its first forward."""
    from mmnn_sts_amd.models.densenet import DenseNet
    ok = dict(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, block_config=(2, 2))
    DenseNet(**ok)
    for bad in (dict(in_channels=5), dict(init_features=96), dict(growth_rate=48), dict(block_config=()), dict(dropout_prob=1.0), dict(bn_size=0)):
        with pytest.raises(ValueError):
            DenseNet(**{**ok, **bad})
    with pytest.raises(NotImplementedError):
        DenseNet(**{**ok, "spatial_dims": 2})
