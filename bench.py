#!/usr/bin/env python
"""Headline benchmark: training volumes/s of the multimodal-fusion step (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = forward of MultiModalModel(DenseNet121-3D(in=2), MLP(32), blend=True) on a micro-batch of 2 synthetic
patients (2 x 2 x 128^3 fp32 volumes + 2 x 32 tabular), GradientBlender-weighted Cox loss, backward, SUM all-reduce of the
gradients over the ranks (N > 1), fused SGD-Nesterov step + OneCycleLR step, zero_grad.  Dropout p = 0.2 active.  Every
rank owns its own patients (weak scaling); value = volumes processed by ALL ranks / max-over-ranks wall time.

Extra legs (rank 0, N = 1 only): `roofline` -- live HIP-event timing of the dominant kernel (3x3x3 dense-layer convolution
forward in dense block 1) against the fp32 MFMA peak; `cpu_baseline` -- the CPU restatement (oracle/, plain torch.nn.functional,
parity-checked against the reference) timed on the host cores for a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_VOLUME = 268.9          # fwd + bwd algorithmic FLOPs of one 2x128^3 volume
PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 MFMA / vector peak (MI355X_MICROARCH.md)
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


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, gfx950 FETCH_SIZE correction applied); None when
    no such measurement is committed.  Counters cannot be read from inside the timed run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes_per_launch"])
    except Exception:
        return None


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


def cpu_baseline(n=2, s=128, steps=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--micro-batch", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    from mmnn_sts_amd import _lib, distributed as D
    from mmnn_sts_amd.losses.GradientBlender import GradientBlender
    from mmnn_sts_amd.losses.losses import CoxPH
    from mmnn_sts_amd.optim import FusedSGD
    from mmnn_sts_amd.utils.utils import surv_criterion

    rank, world, local = D.init_from_env(os.environ.get("MMNN_DIST_BACKEND", "nccl"))   # "nccl" = RCCL; gloo only for rehearsals
    local = local % max(1, torch.cuda.device_count())
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    model = build_model(dev)
    D.broadcast_parameters(model)
    model.train()
    opt = FusedSGD(model, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
    # Initialisation (plan build, lazily created streams/events, allocator growth, clock ramp after the idle model build) takes
    # a handful of steps; they are run before the W warm-up steps when W itself is too small to cover them, and reported.
    init_steps = max(0, 8 - a.warmup)
    total_steps = a.steps + a.warmup + init_steps + 1
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total_steps)
    blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion)
    inputs, events, durations = synth_batch(dev, rank, a.micro_batch, a.size)

    def step():
        out = model(inputs)
        loss, _ = blender.computeLoss(out, events, durations)
        loss.backward()
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
    bb = model.image_model.model.backbone
    plan = next(iter(bb._plans.values()))["plan"]
    L = _lib.lib()
    timed_kernel = rank == 0 and world == 1
    if timed_kernel:
        _lib.check(L.mmnn_densenet_set_timer(plan, 1, 0), "set_timer")    # conv2 (3x3x3) forward, dense block 1
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
        res = {
            "metric": "training volumes/sec/GPU (128^3 T1+T2+preop, batch 2) at 1/2/4/8 MI355X",
            "value": value, "unit": "volumes/s (whole job; 1 volume = 1 patient = stacked T1+T2 2x128^3 + 32 tabular)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "init_steps": init_steps, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "per_gpu": value / world,
            "config": {"workload": "configs[2] --images --preop --survival --blend: MultiModalModel(DenseNet121-3D(in=2), MLP(32), "
                                   "blend) fwd + GradientBlender Cox loss + bwd + grad all-reduce + SGD-Nesterov/OneCycle step, dropout 0.2",
                       "micro_batch": a.micro_batch, "global_batch": a.micro_batch * world, "volume": [2, a.size, a.size, a.size],
                       "tabular": N_CLIN, "parallelism": f"dp{world}"},
            "step_fp32_frac_of_peak": value / world * GFLOP_PER_VOLUME / 1e3 / PEAK_FP32_TFLOPS if a.size == 128 else None,
        }
        if timed_kernel:
            ms, cnt = ctypes.c_double(), ctypes.c_int64()
            _lib.check(L.mmnn_densenet_read_timer(plan, ctypes.byref(ms), ctypes.byref(cnt)), "read_timer")
            L.mmnn_densenet_set_timer(plan, 0, -1)
            v1 = (a.size // 4) ** 3
            flop = 2.0 * a.micro_batch * v1 * 32 * 128 * 27            # one launch: N*V voxels x 32 out x (128 in x 27 taps) MACs x 2
            avg_s = ms.value / max(cnt.value, 1) * 1e-3
            ach = flop / avg_s / 1e12 if avg_s > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "kernel": "fprop_kernel<27,...> conv2 3x3x3 128->32 forward, dense block 1",
                               "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                               "traffic": measured_traffic(), "launches_timed": int(cnt.value), "avg_us": avg_s * 1e6,
                               "flop_per_launch": flop}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.micro_batch, a.size)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
