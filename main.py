#!/usr/bin/env python
"""CLI of the fusion path with the reference's flag surface (main.py:897-1022 of DigITs-AIML/MMNN_STS).

    python main.py --images --preop --survival --blend            # BASELINE config 3 (T1+T2+tabular, GradientBlender)
    python main.py --images --survival                            # config 2 (unimodal DenseNet)
    python main.py --inference --images --preop --survival        # config 5 (Grad-CAM attention maps)
    python main.py --preop --classification                       # config 1 (tabular MLP plumbing)

Training loop = main.py:385-601 restated: micro-batches, gradients accumulated until SUPER_BATCH_SIZE (64) patients were seen,
SGD-Nesterov + OneCycleLR stepped per super-batch, GradientBlender weight update every `--blend_update_interval` epochs,
C-index per epoch, best model (by the un-weighted fused-head loss) saved as best_surv_model.pth.  Deviations from the reference,
all listed in SURVEY Appendix A: the published script cannot be imported (Q1) -- its third assert is dropped; the validation
loop moves `val_images` (Q10); the blender lives on the loss device (Q4); logging syncs once per epoch, not per micro-batch.
Datasets (CSV / NIfTI / DICOM / S3) are host I/O outside the path: without `--data_loc` synthetic patients are used.
With WORLD_SIZE > 1 (torch.distributed.run) patients are sharded over the ranks and gradients SUM-all-reduced (RCCL).
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mmnn_sts_amd import distributed as D  # noqa: E402
from mmnn_sts_amd.data.constants import NUM_CLASSES, SUPER_BATCH_SIZE  # noqa: E402
from mmnn_sts_amd.losses.GradientBlender import GradientBlender  # noqa: E402
from mmnn_sts_amd.losses.losses import CoxPH  # noqa: E402
from mmnn_sts_amd.optim import FusedSGD  # noqa: E402
from mmnn_sts_amd.parser.parser import Parser  # noqa: E402
from mmnn_sts_amd.utils.utils import add_gradcam, loadWeights, surv_criterion  # noqa: E402

logging.basicConfig(level=logging.INFO, format="%(message)s")
logger = logging.getLogger("mmnn_sts_amd")


def str_to_bool(arg):
    if arg.lower() == 'false':
        return False
    if arg.lower() == 'true':
        return True
    raise ValueError('Unexpected value for boolean conversion: {}'.format(arg))


def concordance_index(durations, scores, events):
    """Harrell's C as lifelines.utils.concordance_index(event_times, predicted_scores, event_observed) defines it (main.py:33,122;
    lifelines is not vendored => restated, parity unpinned): over comparable pairs (the earlier time is an observed event),
    a pair is concordant when the higher score goes with the longer time; score ties count 1/2."""
    t, s, e = (np.asarray(a, dtype=np.float64) for a in (durations, scores, events))
    dt = t[:, None] - t[None, :]
    comparable = (dt < 0) & (e[:, None] > 0)
    ds = s[:, None] - s[None, :]
    n = comparable.sum()
    return float(((ds < 0) & comparable).sum() + 0.5 * ((ds == 0) & comparable).sum()) / n if n else float("nan")


def getCIndices(preds, events, durations):
    return [concordance_index(durations[:, i], preds[:, i], events[:, i]) for i in range(NUM_CLASSES)]


class SyntheticPatients(torch.utils.data.Dataset):
    """Stand-in for MultiModalSurvivalDataset (data/MultiModalDatasets.py:8-86): {'image','clinical'}, events, durations."""

    def __init__(self, n, in_channels, size, n_clinical, multimodal, images, seed):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn((n, in_channels, size, size, size), generator=g) if images else None
        self.clinical = torch.randn((n, n_clinical), generator=g)
        self.events = (torch.rand((n, NUM_CLASSES), generator=g) < 0.6).long()
        self.durations = torch.randint(1, 3000, (n, NUM_CLASSES), generator=g)
        self.multimodal = multimodal

    def __len__(self):
        return self.clinical.shape[0]

    def __getitem__(self, i):
        if self.multimodal:
            x = {"image": self.images[i], "clinical": self.clinical[i]}
        else:
            x = self.images[i] if self.images is not None else self.clinical[i]
        return x, self.events[i], self.durations[i]


def collate(batch):
    xs, ev, du = zip(*batch)
    if isinstance(xs[0], dict):
        x = {k: torch.stack([b[k] for b in xs]).float() for k in xs[0]}
    else:
        x = torch.stack(xs).float()
    return x, torch.stack(ev), torch.stack(du)


def to_device(x, device):
    return {k: v.to(device) for k, v in x.items()} if isinstance(x, dict) else x.to(device)


def train_survival(model, train_ds, val_ds, args, device, rank, world):
    loader = torch.utils.data.DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, collate_fn=collate, drop_last=len(train_ds) > args.batch_size)
    val_loader = torch.utils.data.DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, collate_fn=collate, drop_last=len(val_ds) > args.batch_size)
    model = model.to(device)
    D.broadcast_parameters(model)
    super_interval = max(1, SUPER_BATCH_SIZE // (args.batch_size * world))
    steps_per_epoch = max(1, -(-len(loader) // super_interval))
    opt = FusedSGD(model, lr=args.lr, momentum=args.momentum, nesterov=True, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, steps_per_epoch=steps_per_epoch, epochs=args.epochs)
    blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion) if args.blend else None
    best = float("inf")
    for epoch in range(args.epochs):
        model.train()
        losses, c_pred, c_ev, c_du = [], [], [], []
        for i, (x, ev, du) in enumerate(loader):
            x, ev, du = to_device(x, device), ev.to(device), du.to(device)
            out = model(x)
            loss = blender.computeLoss(out, ev, du)[0] if args.blend else surv_criterion(CoxPH, out, ev, du, device)
            loss.backward()
            losses.append(loss.detach())
            if (i + 1) % super_interval == 0 or i == len(loader) - 1:
                D.allreduce_gradients(model)
                opt.step()
                sched.step()
                opt.zero_grad()
            c_pred.append(out.detach()); c_ev.append(ev); c_du.append(du)
        cp = torch.cat(c_pred, dim=1 if args.blend else 0)
        ce, cd = torch.cat(c_ev), torch.cat(c_du)
        fused = cp[0] if args.blend else cp
        tr_c = getCIndices(fused.cpu().numpy(), ce.cpu().numpy(), cd.cpu().numpy())
        model.eval()
        y_pred, y_ev, y_du, sel = [], [], [], None
        with torch.no_grad():
            for x, ev, du in val_loader:
                x, ev, du = to_device(x, device), ev.to(device), du.to(device)
                p = model(x)
                if args.blend:
                    _, sel = blender.computeLoss(p, ev, du)
                else:
                    sel = surv_criterion(CoxPH, p, ev, du, device)
                y_pred.append(p); y_ev.append(ev); y_du.append(du)
        yp = torch.cat(y_pred, dim=1 if args.blend else 0)
        ye, yd = torch.cat(y_ev), torch.cat(y_du)
        val_c = getCIndices((yp[0] if args.blend else yp).cpu().numpy(), ye.cpu().numpy(), yd.cpu().numpy())
        sel = float(sel)
        if rank == 0:
            logger.info(f"epoch {epoch + 1}/{args.epochs} train loss/patient {float(torch.stack(losses).sum()) / len(train_ds):.4f} "
                        f"train C {np.nanmean(tr_c):.3f} val selection loss {sel:.4f} val C {np.nanmean(val_c):.3f}")
            if sel < best:
                best = sel
                os.makedirs(args.output_path, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(args.output_path, 'best_surv_model.pth'))
        if args.blend and (epoch + 1) % args.blend_update_interval == 0:
            blender.updateWeights(cp, ce, cd, yp, ye, yd)
            if rank == 0:
                logger.info('Completed updating gradient blender weights - new weights : {}'.format(blender.weights))
    if args.blend and rank == 0:
        blender.saveHistory()
    return model


def train_classification(model, train_ds, val_ds, args, device):
    """Config 1 plumbing (main.py:125-327 reduced to BCE-with-logits on the event flags): standalone clinical MLP on the device."""
    loader = torch.utils.data.DataLoader(train_ds, batch_size=max(2, args.batch_size), shuffle=True, collate_fn=collate, drop_last=True)
    model = model.to(device)
    opt = torch.optim.SGD(model.parameters(), args.lr, momentum=args.momentum, nesterov=True, weight_decay=args.weight_decay)
    crit = torch.nn.BCEWithLogitsLoss()
    for epoch in range(args.epochs):
        model.train()
        tot = 0.0
        for x, ev, _ in loader:
            out = model(to_device(x, device))
            loss = crit(out, ev.to(device).float())
            loss.backward()
            opt.step(); opt.zero_grad()
            tot += float(loss)
        logger.info(f"epoch {epoch + 1}/{args.epochs} BCE {tot / len(loader):.4f}")
    return model


def inference_survival(model, ds, args, device):
    """main.py:750-887: batch-1 loop, Grad-CAM maps (saved as .npy; NIfTI export needs nibabel, host I/O), C-index."""
    model = model.to(device).eval()
    cam = add_gradcam(model, multimodal=True) if (args.images and args.multimodal and not args.no_gradcam) else None
    preds, evs, dus = [], [], []
    os.makedirs(os.path.join(args.output_path, "attention_maps"), exist_ok=True)
    for i in range(len(ds)):
        x, ev, du = collate([ds[i]])
        x = to_device(x, device)
        with torch.no_grad():
            if cam is not None:
                p, maps = cam(x)
                np.save(os.path.join(args.output_path, "attention_maps", f"patient{i}_att_map.npy"), maps[0].cpu().numpy())
            else:
                p = model(x)
        preds.append(p.cpu()); evs.append(ev); dus.append(du)
    p, e, d = torch.cat(preds).numpy(), torch.cat(evs).numpy(), torch.cat(dus).numpy()
    logger.info('All C-indexes: {}'.format(getCIndices(p, e, d)))
    return p


def main():
    ap = argparse.ArgumentParser()
    for flag, h in (("preop", "clinical features available pre-operation"), ("postop", "pre + post operation clinical features"),
                    ("radiomics", "radiomic features (not implemented upstream either)"), ("images", "image data"),
                    ("classification", "binary classification"), ("survival", "time-to-event model"), ("segmentation", "unsupported"),
                    ("lr_finder", "unsupported tooling"), ("no_gradcam", "disable Grad-CAM for inference"), ("inference", "inference"),
                    ("split", "create a new dataset split"), ("blend", "gradient blending"), ("bootstrap", "bootstrap evaluation")):
        ap.add_argument(f"--{flag}", action="store_true", help=h)
    for twin in ("use_images", "use_preop", "use_postop", "classification_task", "inference_task", "survival_task", "use_blend"):
        ap.add_argument(f"--{twin}", type=str, default="false")
    ap.add_argument("--weights", type=str, default=None)
    ap.add_argument("--output_path", type=str, default=".")
    for loc in ("data_loc", "image_loc", "key_loc", "rad_loc"):
        ap.add_argument(f"--{loc}", type=str, default=None)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--train_uid_location", type=str, default="./stratified_train_uids.txt")
    ap.add_argument("--val_uid_location", type=str, default="./stratified_val_uids.txt")
    ap.add_argument("--config", type=str, default=None)
    ap.add_argument("--blend_update_interval", type=int, default=5)
    # synthetic-data knobs (no counterpart upstream)
    ap.add_argument("--synthetic_patients", type=int, default=16)
    ap.add_argument("--synthetic_size", type=int, default=64)
    a = ap.parse_args()
    a.images = a.images or str_to_bool(a.use_images)
    a.classification = a.classification or str_to_bool(a.classification_task)
    a.inference = a.inference or str_to_bool(a.inference_task)
    a.survival = a.survival or str_to_bool(a.survival_task)
    a.preop = a.preop or str_to_bool(a.use_preop)
    a.postop = a.postop or str_to_bool(a.use_postop)
    a.blend = a.blend or str_to_bool(a.use_blend)
    assert not all([a.classification, a.survival, a.segmentation]), 'Can only specify one of --classification , --survival , or --segmentation'
    assert any([a.classification, a.survival, a.segmentation]), 'Must specify one of --classification , --survival , or --segmentation'
    if a.segmentation or a.lr_finder or a.radiomics:
        raise SystemExit("--segmentation / --lr_finder / --radiomics are outside the MI355X fusion path (SURVEY 2)")
    if a.data_loc or a.image_loc:
        raise SystemExit("real-data loaders (CSV / NIfTI / DICOM / S3) are host I/O outside this path; run without --data_loc for synthetic patients")

    parser = Parser(a.config)
    cfg = parser.parseConfig()
    hp = cfg.get("Hyperparameters", {})
    a.multimodal = a.images and (a.preop or a.postop)
    a.blend = a.blend and a.multimodal
    a.batch_size = int(hp.get("train_batch_size", 2)) if a.config else 2
    a.momentum, a.weight_decay = float(hp.get("momentum", 0.9)), float(hp.get("weight_decay", 1e-4))
    torch.manual_seed(int(hp.get("seed", 42)))
    model = parser.getModel(a)
    if a.multimodal:
        model.blend = a.blend
    rank, world, local = D.init_from_env("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("mmnn_sts_amd runs on the MI355X only (no CPU path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if a.weights:
        model = loadWeights(model, a.weights, "cpu")
    n_clin = len(parser.predictors(a))
    inch = cfg["ImageModel"]["in_channels"]
    mk = lambda n, seed: SyntheticPatients(n, inch, a.synthetic_size, n_clin, a.multimodal, a.images, seed)
    if a.inference:
        inference_survival(model, mk(max(2, a.synthetic_patients // 4), 99), a, device)
    elif a.survival:
        train_survival(model, mk(a.synthetic_patients, 1000 + rank), mk(max(2, a.synthetic_patients // 4), 7), a, device, rank, world)
    else:
        train_classification(model, mk(max(4, a.synthetic_patients), 1000 + rank), None, a, device)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
