"""main.py as a fresh process for the BASELINE configs it serves (tiny synthetic sizes): exit code, saved checkpoint reloads
into a freshly built model with strict=True, attention maps written."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, out):
    env = dict(os.environ, MMNN_POISON_LDS="0", MMNN_POISON_WS="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--output_path", str(out), *args], cwd=str(out), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout + r.stderr


def _tiny_config(tmp_path, modality="t1t2", in_channels=2):
    import yaml
    cfg = {"ImageModel": {"name": "tinydensenet", "modality": modality, "feature_layers": 12, "num_classes": 2, "spatial_dims": 3,
                          "in_channels": in_channels, "dropout_prob": 0.2},
           "ClinicalModel": {"NUM_PREDICTORS": 32, "PRE_OP_PREDICTORS": [], "POST_OP_PREDICTORS": []},
           "Hyperparameters": {"momentum": 0.9, "weight_decay": 1e-4, "train_batch_size": 2, "seed": 42, "class_frequencies": [0.4, 0.55]}}
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(cfg))
    return str(p)


def test_config0_preop_classification(tmp_path):
    """BASELINE configs[0]: `--preop --classification`, 32-feature x 64-patient csv -> standalone MLP, pos-weighted BCE."""
    from mmnn_sts_amd.models.mlp import MLP
    log = _run(["--preop", "--classification", "--epochs", "2", "--synthetic_patients", "64", "--config", _tiny_config(tmp_path)], tmp_path)
    assert "saved new best metric model" in log and "epoch 2/2" in log
    rows = np.loadtxt(tmp_path / "synthetic_train_rank0.csv", delimiter=",", skiprows=1)
    assert rows.shape == (64, 1 + 32 + 4)
    MLP(32, 2, 12).load_state_dict(torch.load(tmp_path / "model.pth"), strict=True)
    MLP(32, 2, 12).load_state_dict(torch.load(tmp_path / "final_model.pth"), strict=True)


def test_config1_images_survival_t1(tmp_path):
    """BASELINE configs[1]: `--images --survival`, single-channel volumes, unimodal DenseNet."""
    from mmnn_sts_amd.models.densenet import TinyDensenet
    log = _run(["--images", "--survival", "--epochs", "2", "--synthetic_patients", "6", "--synthetic_size", "32",
                "--config", _tiny_config(tmp_path, "t1", 1)], tmp_path)
    assert "epoch 2/2" in log
    m = TinyDensenet(spatial_dims=3, in_channels=1, out_channels=2, feature_channels=12, dropout_prob=0.2)
    m.load_state_dict(torch.load(tmp_path / "best_surv_model.pth"), strict=True)


def test_config2_fusion_blend_then_config4_gradcam(tmp_path):
    """BASELINE configs[2] then configs[4]: train the fusion model with the GradientBlender (weights updated after epoch 2), then
    reload the checkpoint for Grad-CAM inference."""
    from mmnn_sts_amd.models.densenet import TinyDensenet
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    cfg = _tiny_config(tmp_path)
    log = _run(["--images", "--preop", "--survival", "--blend", "--blend_update_interval", "2", "--epochs", "2", "--synthetic_patients", "6",
                "--synthetic_size", "32", "--config", cfg], tmp_path)
    assert "Completed updating gradient blender weights" in log
    hist = np.loadtxt(tmp_path / "gblend_weights_history.csv", delimiter=",", ndmin=2)
    assert hist.shape == (1, 3) and abs(hist.sum() - 1.0) < 1e-5
    img = TinyDensenet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=0.2)
    MultiModalModel(img, [f"p{i}" for i in range(32)], 2, 12, blend=True).load_state_dict(torch.load(tmp_path / "best_surv_model.pth"), strict=True)
    log = _run(["--inference", "--images", "--preop", "--survival", "--weights", str(tmp_path / "best_surv_model.pth"), "--synthetic_patients", "8",
                "--synthetic_size", "32", "--config", cfg], tmp_path)
    assert "All C-indexes" in log
    m = np.load(tmp_path / "attention_maps" / "patient0_att_map.npy")
    assert m.shape == (32, 32, 32) and np.isfinite(m).all() and m.min() >= 0.0 and m.max() <= 1.0 + 1e-6


def test_bench_two_ranks_at_baseline_extent(tmp_path):
    """BASELINE configs[3] rehearsal on ONE card: `bench.py --gpus 2 --size 128` starts its two ranks itself, the
    ranks share the card over gloo (RCCL needs one device per rank), gradients are all-reduced per dense block from inside the backward.
    The JSON line must report both ranks, a finite throughput and the overlapped schedule."""
    import json
    env = dict(os.environ, MMNN_DIST_BACKEND="gloo", MMNN_POISON_LDS="0", MMNN_POISON_WS="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "128", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == d["ranks_seen"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["config"]["allreduce"].startswith("per dense block") and d["scaling"] == "weak"
    assert np.isfinite(d["value"]) and d["value"] > 0 and d["steps"] == 2


def _run_ranks(args, out, world=2):
    """main.py under torch.distributed.run, `world` ranks sharing the one card over gloo (on an 8-GPU node the same command runs over RCCL)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MMNN_DIST_BACKEND="gloo", MMNN_POISON_LDS="0", MMNN_POISON_WS="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "main.py"), "--output_path", str(out), *args], cwd=str(out), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout + r.stderr


def test_two_rank_survival_blend_and_classification(tmp_path):
    """The CLI's data-parallel paths on two ranks: survival + GradientBlender (accumulation windows whose last backward arms the overlapped
    reducer, blender update on gathered predictions, rank 0 saves) and classification (broadcast, all-reduce before every
    optimizer step, checkpoints written by rank 0 only)."""
    from mmnn_sts_amd.models.densenet import TinyDensenet
    from mmnn_sts_amd.models.mlp import MLP
    from mmnn_sts_amd.models.multimodal import MultiModalModel
    cfg = _tiny_config(tmp_path)
    a = tmp_path / "surv"; a.mkdir()
    log = _run_ranks(["--images", "--preop", "--survival", "--blend", "--blend_update_interval", "1", "--epochs", "1", "--synthetic_patients", "6",
                      "--synthetic_size", "32", "--config", cfg], a)
    assert "Completed updating gradient blender weights" in log and log.count("saved new best metric model") == 1      # rank 0 only
    img = TinyDensenet(spatial_dims=3, in_channels=2, out_channels=2, feature_channels=12, dropout_prob=0.2)
    MultiModalModel(img, [f"p{i}" for i in range(32)], 2, 12, blend=True).load_state_dict(torch.load(a / "best_surv_model.pth"), strict=True)
    b = tmp_path / "cls"; b.mkdir()
    log = _run_ranks(["--preop", "--classification", "--epochs", "2", "--synthetic_patients", "16", "--config", cfg], b)
    assert log.count("epoch 2/2") == 1                                                                                  # rank 0 logs
    MLP(32, 2, 12).load_state_dict(torch.load(b / "final_model.pth"), strict=True)
    assert sorted(p.name for p in b.glob("synthetic_train_rank*.csv")) == ["synthetic_train_rank0.csv", "synthetic_train_rank1.csv"]
