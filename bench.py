#!/usr/bin/env python
"""Headline benchmark: training volumes/s of the multimodal-fusion step (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = forward of MultiModalModel(DenseNet121-3D(in=2), MLP(32), blend=True) on a micro-batch of 2 synthetic
patients (2 x 2 x 128^3 fp32 volumes + 2 x 32 tabular), GradientBlender-weighted Cox loss, backward, SUM all-reduce of the
gradients over the ranks (N > 1), fused SGD-Nesterov step + OneCycleLR step, zero_grad.  Dropout p = 0.2 active.  Every
rank owns its own patients (weak scaling); value = volumes processed by ALL ranks / max-over-ranks wall time.

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (torch.distributed.run as a child process, before
anything here touches the GPU) and relays rank 0's JSON line; every rank asserts that the process group really has N members.

`--config unimodal` times BASELINE configs[1] (DenseNet121(in=1) + surv_criterion(CoxPH), 2 x 1 x 128^3) and `--config gradcam256`
configs[4] (Grad-CAM inference of one 1 x 2 x 256^3 patient: eval forward + mmnn_gradcam, s/patient) the same way; they print a JSON line
of their own metric (the headline line is the default `--config fusion`).

Extra legs (rank 0, N = 1 only): `roofline` -- after the timed region, a few more (untimed) steps with every convolution launch
bracketed by HIP events on its own launch stream and the backward serialised on one stream (un-overlapped durations); the
kernel class with the largest total time is reported against the fp32 MFMA peak, the other classes beside it;
`cpu_baseline` -- the CPU restatement (oracle/, plain torch.nn.functional, parity-checked against the reference) timed on
the host cores for a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_VOLUME = 268.9          # fwd + bwd algorithmic FLOPs of one 2x128^3 volume
PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 MFMA / vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0         # dense bf16 MFMA peak (MI355X_MICROARCH.md; the headline figures with 2:1 sparsity are not used)
N_CLIN = 32


def build_model(device, blend=True, dropout=0.2):
    from mmnn_sts_amd.models.densenet import DenseNet121
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    torch.manual_seed(42)
    img = DenseNet121(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=dropout)
    return MultiModalModel(img, [f"p{i}" for i in range(N_CLIN)], 2, 12, blend=blend).to(device)


def synth_batch(device, rank, n=2, s=128):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    image = torch.randn((n, 2, s, s, s), device=device, generator=g)
    clinical = torch.randn((n, N_CLIN), device=device, generator=g)
    events = (torch.rand((n, 2), device=device, generator=g) < 0.5).long()
    events[0] = 1
    durations = torch.randint(1, 3000, (n, 2), device=device, generator=g)
    return {"image": image, "clinical": clinical}, events, durations


def measured_traffic(key=None):
    """HBM bytes per launch of a kernel class from the committed PMC passes (profiles/rNN_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, gfx950 FETCH_SIZE correction applied); None when
    no such measurement is committed.  Counters cannot be read from inside the timed run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        if key is not None and isinstance(d.get("kernels"), dict):
            e = d["kernels"].get(key)
            return float(e["hbm_bytes_per_launch"]) if e else None
        return float(d["hbm_bytes_per_launch"]) if key in (None, "conv2_fwd.b1") else None
    except Exception:
        return None


KCLASS = {1: "conv2_fwd", 2: "conv2_dgrad", 3: "conv2_wgrad", 4: "conv1_fwd", 5: "conv1_dgrad", 6: "conv1_wgrad", 7: "stem_conv",
          8: "stem_wgrad", 9: "block_fwd"}
KDESC = {1: "fprop_kernel<27,PRO_BNRELU,EPI_STORE_STATS> conv2 3x3x3 128->32 forward",
         2: "fprop_kernel<27,PRO_GRAD,EPI_MASK_STORE> conv2 3x3x3 data gradient 32->128",
         3: "wgrad3_kernel conv2 3x3x3 weight gradient",
         4: "fprop_kernel<1,PRO_BNRELU,EPI_STORE_STATS> conv1 1x1x1 C->128 forward",
         5: "fprop_kernel<1,PRO_GRAD,EPI_MASK_ACCUM> conv1 1x1x1 data gradient 128->C",
         6: "wgrad1_kernel conv1 1x1x1 weight gradient",
         7: "stem_conv_kernel conv0 7x7x7 stride 2 forward", 8: "stem_wgrad_kernel conv0 weight gradient",
         9: "block_fwd_kernel: conv1 + conv2 forward of EVERY layer of the dense block, one persistent launch"}


def class_flops(kind, block, n, size, in_ch=2, blocks=(6, 12, 24, 16), growth=32, mid=128, init=64):
    """Algorithmic FLOPs of ALL launches of one kernel class in one dense block during one step (2 x MACs)."""
    if kind in (7, 8):
        return 2.0 * n * (size // 2) ** 3 * init * in_ch * 343
    v = (size // 4) ** 3
    c = init
    for b in range(block):
        c = (c + blocks[b] * growth) // 2
        v //= 8
    conv2 = blocks[block] * 2.0 * n * v * growth * mid * 27
    conv1 = sum(2.0 * n * v * mid * (c + growth * l) for l in range(blocks[block]))
    if kind == 9:
        return conv1 + conv2
    return conv2 if kind in (1, 2, 3) else conv1


HBM_ACHIEVABLE_TBS = 6.29         # measured copy rate (MI355X_MICROARCH.md HBM section; spec 8.0)
HBM_SPEC_TBS = 8.0


def class_bytes(kind, block, n, size, in_ch=2, blocks=(6, 12, 24, 16), growth=32, mid=128, init=64):
    """ALGORITHMIC HBM bytes of ALL launches of one kernel class in one dense block during one step: every operand tensor read once,
    every result written once (SURVEY 8(d) "fused minimum"; weight-gradient partial slabs are not algorithmic and not counted)."""
    f = 4.0
    if kind == 7:      # conv0: read x, write conv0 output (+ weights)
        return f * (n * in_ch * size ** 3 + n * init * (size // 2) ** 3 + init * in_ch * 343)
    if kind == 8:      # conv0 weight gradient: read x, dZ0 and conv0's output (BN backward on operand load), write dW
        return f * (n * in_ch * size ** 3 + 2 * n * init * (size // 2) ** 3 + init * in_ch * 343)
    v = (size // 4) ** 3
    c = init
    for b in range(block):
        c = (c + blocks[b] * growth) // 2
        v //= 8
    nv = n * v
    w2 = 27 * growth * mid
    tot = 0.0
    for l in range(blocks[block]):
        cin = c + growth * l
        w1 = mid * cin
        if kind == 9:   tot += nv * (cin + mid) + w1 + nv * (mid + growth) + w2   # both forward convolutions of the layer
        elif kind == 1: tot += nv * (mid + growth) + w2                      # read T1, write the new channels
        elif kind == 2: tot += nv * (2 * growth + 2 * mid) + w2              # read G and X slices, read T1 (mask), write dZ2
        elif kind == 3: tot += nv * (2 * growth + mid) + w2                  # read G and X slices, read T1, write dW
        elif kind == 4: tot += nv * (cin + mid) + w1                         # read the concat, write T1
        elif kind == 5: tot += nv * (2 * mid + 3 * cin) + w1                 # read dZ2 and T1, read X (mask), read + write G
        elif kind == 6: tot += nv * (2 * mid + cin) + w1                     # read dZ2 and T1, read X, write dW
    return f * tot


def kernel_roofline(L, plan, step, n, size, steps=4, in_ch=2):
    """Per-class device time of every convolution launch: events on the launch stream, backward on ONE stream so that the
    durations are not inflated by kernels overlapping on the side streams.  Runs `steps` extra steps after the timed region."""
    _lib_check = __import__("mmnn_sts_amd._lib", fromlist=["check"]).check
    _lib_check(L.mmnn_densenet_set_option(plan, b"single_stream", 1), "set_option")
    step()                                                     # settle into the serialised schedule
    torch.cuda.synchronize()
    _lib_check(L.mmnn_densenet_set_timer(plan, -1, -1), "set_timer")
    rows = {}
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
        for kind in KCLASS:
            for b in range(1 if kind in (7, 8) else 4):
                ms, cnt = ctypes.c_double(), ctypes.c_int64()
                _lib_check(L.mmnn_densenet_read_timer_class(plan, kind, b, ctypes.byref(ms), ctypes.byref(cnt)), "read_timer")
                rows[(kind, b)] = (ms.value, cnt.value)     # accumulated since set_timer
    L.mmnn_densenet_set_timer(plan, 0, -1)
    L.mmnn_densenet_set_option(plan, b"single_stream", 0)
    out = []
    for (kind, b), (ms, cnt) in rows.items():
        if cnt == 0:
            continue
        flop = class_flops(kind, b, n, size, in_ch=in_ch) * steps
        byts = class_bytes(kind, b, n, size, in_ch=in_ch) * steps
        tf = flop / (ms * 1e-3) / 1e12
        tbs = byts / (ms * 1e-3) / 1e12
        name = KCLASS[kind] + ("" if kind in (7, 8) else f".b{b + 1}")
        # conv2 forward and data gradient at extents wider than 16 voxels run on the bf16 matrix pipe with three-piece operands (csrc/conv3_bf16x3.hip):
        # six bf16 MFMA products per fp32 product, priced against the dense bf16 peak
        bf16x3_mode = os.environ.get("MMNN_BF16X3", "32")
        wide = ((size // 4) >> b) > 16
        bf16x3 = bf16x3_mode != "0" and ((kind == 1 and (wide or (bf16x3_mode == "16" and ((size // 4) >> b) > 8)))
                                         or (kind == 2 and wide and os.environ.get("MMNN_BF16X3_DGRAD", "1") != "0"))
        peak_tf, pipe_mult = (PEAK_BF16_TFLOPS, 6.0) if bf16x3 else (PEAK_FP32_TFLOPS, 1.0)
        # which roof bounds the class: the one its algorithmic work takes longer to cross
        hbm_bound = byts / (HBM_ACHIEVABLE_TBS * 1e12) > pipe_mult * flop / (peak_tf * 1e12)
        row = {"class": name, "kernel": KDESC[kind] + ("" if kind in (7, 8) else f", dense block {b + 1}"), "launches": int(cnt),
               "ms_per_step": ms / steps, "avg_us": ms / cnt * 1e3, "flop_per_launch": flop / cnt, "bytes_per_launch": byts / cnt,
               "bound": "hbm" if hbm_bound else "mfma", "mfma_tflops": pipe_mult * tf, "mfma_frac": pipe_mult * tf / peak_tf, "hbm_tbs": tbs,
               "hbm_frac": tbs / HBM_ACHIEVABLE_TBS, "hbm_frac_of_spec": tbs / HBM_SPEC_TBS}
        if bf16x3:
            row["pipe"] = "bf16, three pieces per fp32 operand: 6 MFMA products per fp32 product (mfma_tflops counts them; peak = dense bf16)"
            row["fp32_equivalent_tflops"] = tf
        row["achieved"], row["peak"], row["unit"] = (tbs, HBM_ACHIEVABLE_TBS, "TB/s") if hbm_bound else (pipe_mult * tf, peak_tf, "TFLOP/s")
        row["frac"] = row["achieved"] / row["peak"]
        if max(row["mfma_frac"], row["hbm_frac"]) < 0.15:
            row["limiter"] = "latency"      # neither roof is near: a chain of memory round trips (the 8^3 / 4^3 layers)
        out.append(row)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a child torch.distributed.run, relay its
    output and exit code.  Nothing in this process has touched the GPU yet (no HIP call, no torch.cuda.is_available())."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def host_cores() -> int:
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box gives one GPU's
    share of the host, not every core `os.cpu_count()` reports)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(n=2, s=128, steps=5):
    """The oracle's plain-torch CPU path (kind "port"), same workload, bounded sample: 1 warm-up + `steps` timed steps."""
    from oracle import restatement as R
    from oracle import synth
    import numpy as np
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = R.DenseNetCfg(in_channels=2, dropout_prob=0.2)
    sch = R.multimodal_schema(cfg, N_CLIN, 2, 12)
    sd = {}
    for k, v in synth.synth_state_dict(sch, "fusion.").items():
        t = torch.from_numpy(np.asarray(v))
        sd[k] = t.requires_grad_(True) if (t.is_floating_point() and "running" not in k) else t
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.SGD(params, 1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    image = torch.randn((n, 2, s, s, s), generator=g)
    clinical = torch.randn((n, N_CLIN), generator=g)
    events = (torch.rand((n, 2), generator=g) < 0.5).long()
    events[0] = 1
    durations = torch.randint(1, 3000, (n, 2), generator=g)
    blender = R.Blender()
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        out = R.multimodal_forward(sd, image, clinical, cfg, True, True, mlp_dropout=0.2)
        loss, _ = blender.compute_loss(out, events, durations)
        loss.backward()
        opt.step()
        opt.zero_grad()
        if i:
            times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {i}: {time.perf_counter() - t0:.2f} s ({cores} threads)", file=sys.stderr, flush=True)
    sec = sum(times) / len(times)
    return {"value": n / sec, "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the same micro-batch-2 128^3 step, {sec:.2f} s/step, torch {torch.__version__} CPU"}


def measure_clock(L, dev):
    """MHz sustained under a chip-wide fp32 MFMA load on THIS box (csrc/probe.hip): explains box-to-box differences of the fractions."""
    mhz = ctypes.c_double()
    scratch = torch.zeros(4, device=dev)
    rc = L.mmnn_measure_mfma_clock(ctypes.byref(mhz), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return float(mhz.value) if rc == 0 else None


def run_gradcam256(a, dev, L):
    """BASELINE configs[4]: Grad-CAM attention maps of one 1 x 2 x S^3 patient (S = 256): eval-mode HIP forward + mmnn_gradcam."""
    model = build_model(dev, blend=False).eval()
    cam = model.add_gradcam("unused")
    s = a.size
    g = torch.Generator(device=dev).manual_seed(99)
    x = {"image": torch.randn((1, 2, s, s, s), device=dev, generator=g), "clinical": torch.randn((1, N_CLIN), device=dev, generator=g)}
    for _ in range(max(2, a.warmup)):
        cam(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        preds, maps = cam(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # device time of the Grad-CAM arithmetic alone (everything after the model forward): events around the C-ABI call's kernels
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd_ms = []
    for _ in range(3):
        e0.record()
        with torch.no_grad():
            model.image_model.model.backbone(x["image"])
        e1.record()
        torch.cuda.synchronize()
        fwd_ms.append(e0.elapsed_time(e1))
    fwd = sorted(fwd_ms)[1]
    out_bytes = 4.0 * len(maps) * s ** 3
    flops = 778.5e9 * (s / 256.0) ** 3                        # eval forward of one 2 x 256^3 volume
    assert torch.isfinite(preds).all() and all(torch.isfinite(m).all() for m in maps)
    return {"metric": "Grad-CAM inference seconds/patient (1 x 2 x %d^3, configs[4])" % s, "value": dt, "unit": "s/patient",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": False, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4] --inference --images --preop --survival: eval forward of MultiModalModel(DenseNet121-3D(in=2), MLP(32)) "
                                   "+ mmnn_gradcam (closed-form gradient, pooled weighting, min-max, trilinear up-sampling of 2 maps)",
                       "volume": [2, s, s, s], "tabular": N_CLIN, "batch": 1},
            "roofline": {"bound": "mfma", "kernel": "eval-mode backbone forward (all convolutions)", "achieved": flops / (fwd * 1e-3) / 1e12,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": flops / (fwd * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "traffic": None,
                         "note": "fp32-equivalent FLOPs of the whole eval forward against the fp32 MFMA peak; the conv2 layers of the 64^3 and 32^3 blocks run "
                                 "on the bf16 pipe with three-piece operands (csrc/conv3_bf16x3.hip), so this is a throughput figure, not one kernel's roofline",
                         "backbone_forward_ms": fwd, "gradcam_tail_ms": dt * 1e3 - fwd,
                         "upsample_bytes_written": out_bytes, "clock_mhz": measure_clock(L, dev)},
            "reference_cpu_s_per_patient": "23-36 (BASELINE.md, reference on 8 vCPU)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--micro-batch", type=int, default=2)
    ap.add_argument("--config", choices=("fusion", "unimodal", "gradcam256"), default="fusion")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-reduce after the whole backward instead of per dense block")
    a = ap.parse_args()
    if a.size is None:
        a.size = 256 if a.config == "gradcam256" else 128
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a))

    from mmnn_sts_amd import _lib, distributed as D
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    from mmnn_sts_amd.losses.losses import CoxPH
    from mmnn_sts_amd.optim import FusedSGD
    from mmnn_sts_amd.utils.utils import surv_criterion

    rank, world, local = D.init_from_env(os.environ.get("MMNN_DIST_BACKEND", "nccl"))   # "nccl" = RCCL; gloo only for rehearsals
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: every rank must be launched (torch.distributed.run or plain `bench.py --gpus N`)")
    ndev = torch.cuda.device_count()
    if ndev < world and os.environ.get("MMNN_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {a.gpus} but only {ndev} device(s) visible (ranks may share a device only in gloo rehearsals)")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = _lib.lib()
    if a.config == "gradcam256":
        if world != 1:
            raise SystemExit("--config gradcam256 is a single-GPU inference benchmark")
        print(json.dumps(run_gradcam256(a, dev, L)), flush=True)
        return
    ranks_seen = 1
    if world > 1:
        assert torch.distributed.get_world_size() == a.gpus
        seen = torch.ones(1, device=dev)
        torch.distributed.all_reduce(seen)
        ranks_seen = int(seen.item())
        assert ranks_seen == a.gpus, f"{ranks_seen} ranks answered the all-reduce, expected {a.gpus}"
    unimodal = a.config == "unimodal"
    in_ch = 1 if unimodal else 2
    if unimodal:
        from mmnn_sts_amd.models.densenet import DenseNet121
        torch.manual_seed(42)
        model = DenseNet121(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, dropout_prob=0.2).to(dev)
    else:
        model = build_model(dev)
    D.broadcast_parameters(model)
    model.train()
    opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
    # Initialisation (plan build, lazily created streams/events, allocator growth, clock ramp after the idle model build) takes
    # a handful of steps; they are run before the W warm-up steps when W itself is too small to cover them, and reported.
    init_steps = max(0, 8 - a.warmup)
    total_steps = a.steps + a.warmup + init_steps + 1 + 8      # + the untimed steps of the roofline leg
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total_steps)
    blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
    inputs, events, durations = synth_batch(dev, rank, a.micro_batch, a.size)
    if unimodal:
        inputs = inputs["image"][:, :1].contiguous()
    # N > 1: the gradient all-reduce is issued per dense block from inside the backward (block 4 + norm5 first) and overlaps the
    # kernels of the remaining blocks; `--no-overlap` reduces everything after the backward instead (same result bit for bit)
    reducer = D.OverlappedGradientReducer(model) if (world > 1 and not a.no_overlap) else None

    def step():
        out = model(inputs)
        loss = surv_criterion(CoxPH, out, events, durations, dev) if unimodal else blender.computeLoss(out, events, durations)[0]
        if reducer is not None:
            reducer.arm()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        else:
            D.allreduce_gradients(model)
        opt.step()
        sched.step()
        opt.zero_grad()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(init_steps + a.warmup):
        step()
    bb = model.backbone if unimodal else model.image_model.model.backbone
    plan = next(iter(bb._plans.values()))["plan"]
    timed_kernel = rank == 0 and world == 1 and not a.no_roofline
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        vols = a.steps * a.micro_batch * world
        value = vols / dt
        gflop_vol = 245.9 if unimodal else GFLOP_PER_VOLUME      # in=1 / in=2
        if unimodal:
            metric = "training volumes/sec/GPU (128^3 T1, batch 2, configs[1]) on 1 MI355X"
            unit = "volumes/s (1 volume = 1 patient = one T1 1x128^3 volume)"
            workload = ("configs[1] --images --survival, modality t1: DenseNet121-3D(in=1) fwd + surv_criterion(CoxPH) + bwd + grad all-reduce + "
                        "SGD-Nesterov/OneCycle step, dropout 0.2")
        else:
            metric = "training volumes/sec/GPU (128^3 T1+T2+preop, batch 2) at 1/2/4/8 MI355X"
            unit = "volumes/s (whole job; 1 volume = 1 patient = stacked T1+T2 2x128^3 + 32 tabular)"
            workload = ("configs[2] --images --preop --survival --blend: MultiModalModel(DenseNet121-3D(in=2), MLP(32), "
                        "blend) fwd + GradientBlender Cox loss + bwd + grad all-reduce + SGD-Nesterov/OneCycle step, dropout 0.2")
        res = {
            "metric": metric, "value": value, "unit": unit,
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup, "init_steps": init_steps, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 tensors and fp32 accumulation throughout; the dense-layer conv2 forward and data gradient at extents wider than 16 voxels multiply "
                          "three-piece bf16 splits of their fp32 operands (6 bf16 MFMA products per fp32 product, error at the fp32 MFMA's own level; "
                          "every parity test runs at unchanged tolerance; MMNN_BF16X3=0 restores the fp32 MFMA kernel)",
            "per_gpu": value / world,
            "config": {"workload": workload, "micro_batch": a.micro_batch, "global_batch": a.micro_batch * world,
                       "volume": [in_ch, a.size, a.size, a.size], "tabular": 0 if unimodal else N_CLIN, "parallelism": f"dp{world}",
                       "allreduce": ("none" if world == 1 else ("per dense block, overlapped with the backward" if reducer is not None else "after the backward"))},
            "step_fp32_frac_of_peak": value / world * gflop_vol / 1e3 / PEAK_FP32_TFLOPS if a.size == 128 else None,
        }
        if timed_kernel:
            rows = kernel_roofline(L, plan, step, a.micro_batch, a.size, in_ch=in_ch)
            top = rows[0]
            clock = measure_clock(L, dev)
            res["roofline"] = {"bound": top["bound"], "kernel": top["kernel"], "class": top["class"], "achieved": top["achieved"],
                               "peak": top["peak"], "unit": top["unit"], "frac": top["frac"], "traffic": measured_traffic(top["class"]),
                               "launches_timed": top["launches"], "avg_us": top["avg_us"], "flop_per_launch": top["flop_per_launch"],
                               "bytes_per_launch": top["bytes_per_launch"],
                               "ms_per_step": top["ms_per_step"], "step_frac": res["step_fp32_frac_of_peak"],
                               "clock_mhz": clock, "nominal_clock_mhz": 2400.0,
                               "frac_at_measured_clock": (top["frac"] * 2400.0 / clock) if (clock and top["bound"] == "mfma") else None,
                               "timing": "HIP events on the launch stream, 4 untimed steps after the timed region, backward on one stream",
                               "bounds": "per class: 'mfma' (fraction of 157.3 TFLOP/s fp32) or 'hbm' (algorithmic bytes / time against 6.29 TB/s "
                                         "measured copy rate, 8.0 spec), whichever roof its algorithmic work takes longer to cross; 'limiter: latency' "
                                         "where neither fraction reaches 0.15",
                               "conv_ms_per_step": sum(r["ms_per_step"] for r in rows),
                               "small_extent_ms_per_step": sum(r["ms_per_step"] for r in rows if r["class"].endswith((".b3", ".b4"))),
                               "classes": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "kernel"} for r in rows]}
        if world == 1 and not a.no_cpu_baseline and not unimodal:
            res["cpu_baseline"] = cpu_baseline(a.micro_batch, a.size)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
