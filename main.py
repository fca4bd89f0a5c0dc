#!/usr/bin/env python
"""CLI of the fusion path with the reference's flag surface (main.py:897-1022 of DigITs-AIML/MMNN_STS).

    python main.py --preop --classification                       # BASELINE configs[0] (tabular MLP from a 32 x 64 csv)
    python main.py --images --survival                            # configs[1] (unimodal DenseNet, t1)
    python main.py --images --preop --survival --blend            # configs[2] (T1+T2+tabular, GradientBlender)
    python main.py --inference --images --preop --survival        # configs[4] (Grad-CAM attention maps)

Training loops = main.py:385-601 (survival) and :125-327 (classification) restated.  Survival: micro-batches, gradients
accumulated until SUPER_BATCH_SIZE (64) patients were seen, SGD-Nesterov + OneCycleLR stepped per super-batch, GradientBlender
weight update every `--blend_update_interval` epochs, C-index per epoch, best model (by the un-weighted fused-head loss of the
last validation batch, as upstream :525,573) saved as best_surv_model.pth.  Classification: pos-weighted BCE on logits, one
optimizer step per batch, F1 per epoch, best model by mean validation F1 saved as model.pth, optional GradientBlender.
Deviations from the reference, all listed in SURVEY Appendix A: the published script cannot be imported (Q1) -- its third assert
is dropped and CLASS_FREQUENCIES comes from the config; the validation loop moves `val_images` (Q10); the blender lives on the
loss device (Q4); logging syncs once per epoch, not per micro-batch; `--preop` alone builds the standalone MLP (Q12).
Datasets (NIfTI / DICOM / S3) are host I/O outside the path: without `--data_loc` synthetic patients are used; the tabular-only
config reads them back from a csv it writes first (the "synthetic 32-feature x 64-patient csv" of BASELINE configs[0]).
There is no CPU compute path: every model runs on the MI355X through the HIP library (configs[0]'s "CPU" is upstream's device).
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
from mmnn_sts_amd.data.constants import CLASSIFICATION_THRESHOLD, NUM_BOOTSTRAP_ITERATIONS, NUM_CLASSES, SUPER_BATCH_SIZE  # noqa: E402
from mmnn_sts_amd.losses.GradientBlender import GradientBlender  # noqa: E402
from mmnn_sts_amd.losses.losses import BCEWithLogitsLoss, CoxPH  # noqa: E402
from mmnn_sts_amd.optim import FusedSGD  # noqa: E402
from mmnn_sts_amd.parser.parser import Parser  # noqa: E402
from mmnn_sts_amd.utils.utils import add_gradcam, criterion, loadWeights, surv_criterion  # noqa: E402

logging.basicConfig(level=logging.INFO, format="%(message)s")
logger = logging.getLogger("mmnn_sts_amd")


def str_to_bool(arg):
    if arg.lower() == 'false':
        return False
    if arg.lower() == 'true':
        return True
    raise ValueError('Unexpected value for boolean conversion: {}'.format(arg))


# ---- epoch-level bookkeeping (host logic, unit-tested on CPU) --------------------------------------------------------------
def concordance_index(event_times, predicted_scores, event_observed):
    """Harrell's C as `lifelines.utils.concordance_index(event_times, predicted_scores, event_observed)` computes it (main.py:33,
    122; lifelines is not vendored => restated from its published algorithm, parity unpinned).  An ordered pair (a, b) is
    admissible when a is an observed event and b outlived a: t_b > t_a, or t_b == t_a with b censored (lifelines handles the
    deaths of a time before its censored cases).  It is concordant when a also has the lower score, half-credit for equal scores.
    NaN when no pair is admissible (lifelines raises ZeroDivisionError there)."""
    t, s, e = (np.asarray(a, dtype=np.float64).reshape(-1) for a in (event_times, predicted_scores, event_observed))
    died = e > 0
    later = t[None, :] > t[:, None]
    same_time_censored = (t[None, :] == t[:, None]) & ~died[None, :]
    admissible = died[:, None] & (later | same_time_censored)
    n = int(admissible.sum())
    if n == 0:
        return float("nan")
    lower = s[:, None] < s[None, :]
    equal = s[:, None] == s[None, :]
    return float((lower & admissible).sum() + 0.5 * (equal & admissible).sum()) / n


def getCIndices(preds, events, durations):
    """main.py:106-123: one C-index per survival target."""
    return [concordance_index(durations[:, i], preds[:, i], events[:, i]) for i in range(NUM_CLASSES)]


def getF1Score(tps, fps, fns):
    """main.py:97-104."""
    return [float(tps[i] / (tps[i] + 0.5 * (fns[i] + fps[i]))) for i in range(NUM_CLASSES)]


def super_batch_interval(batch_size: int, world: int = 1) -> int:
    """Micro-batches per optimizer step (main.py:403): gradients are summed until SUPER_BATCH_SIZE patients were seen -- by all
    ranks together when the patients are sharded over `world` ranks."""
    return max(1, SUPER_BATCH_SIZE // (batch_size * world))


def optimizer_steps_per_epoch(n_batches: int, interval: int) -> int:
    """Optimizer / scheduler steps of one epoch: every `interval`-th micro-batch plus the ragged tail.  Equals upstream's
    ceil(len(train_dataset) / SUPER_BATCH_SIZE) (main.py:404-407) whenever the batch size divides SUPER_BATCH_SIZE."""
    return max(1, -(-n_batches // interval))


def is_step_boundary(i: int, n_batches: int, interval: int) -> bool:
    """main.py:478: step on every `interval`-th micro-batch and on the last one of the epoch."""
    return (i + 1) % interval == 0 or i == n_batches - 1


def blender_update_due(epoch: int, interval: int) -> bool:
    """main.py:584: epochs are 0-based here as upstream."""
    return (epoch + 1) % interval == 0


def gather_rows(t: torch.Tensor, dim: int, world: int) -> torch.Tensor:
    """Concatenate a per-rank tensor over all ranks along `dim` (identity for one rank): the blender's epoch-level losses must be
    computed on ALL patients so that every rank derives the same head weights."""
    if world == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(world)]
    torch.distributed.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=dim)


def bootstrap_c_indices(preds, events, durations, iterations: int = NUM_BOOTSTRAP_ITERATIONS, seed: int = 0):
    """main.py:767-768,857-887 (`--bootstrap`): C-indices of `iterations` resamples (with replacement, same size) of the evaluated
    patients -> (per-target means, per-target standard deviations, number of usable resamples).  Upstream re-runs the model over every
    resampled uid list; in eval mode with batch size 1 a patient's prediction does not depend on its neighbours, so resampling the
    PREDICTIONS of one pass gives the same numbers at 1/50th of the device work.  A resample without any admissible pair is skipped
    (upstream: `except ZeroDivisionError: continue`, :857-858)."""
    preds, events, durations = (np.asarray(t) for t in (preds, events, durations))
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(iterations):
        idx = rng.integers(0, len(preds), len(preds))
        c = getCIndices(preds[idx], events[idx], durations[idx])
        if not np.any(np.isnan(c)):
            rows.append(c)
    if not rows:
        return [float("nan")] * NUM_CLASSES, [float("nan")] * NUM_CLASSES, 0
    arr = np.asarray(rows)
    return arr.mean(axis=0).tolist(), arr.std(axis=0).tolist(), len(rows)


# ---- data plumbing -----------------------------------------------------------------------------------------------------------
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


def write_synthetic_csv(path, n_patients, predictors, seed):
    """The tabular plumbing of BASELINE configs[0]: one row per patient -- uid, the predictor columns, then per target an event
    flag and a duration -- as data/ClinicalDatasets.py:6-89 reads it from its csv."""
    g = np.random.default_rng(seed)
    cols = ["uid"] + list(predictors) + [f"event{i}" for i in range(NUM_CLASSES)] + [f"duration{i}" for i in range(NUM_CLASSES)]
    rows = np.concatenate([np.arange(n_patients)[:, None], g.standard_normal((n_patients, len(predictors))),
                           (g.random((n_patients, NUM_CLASSES)) < 0.6).astype(np.float64),
                           g.integers(1, 3000, (n_patients, NUM_CLASSES)).astype(np.float64)], axis=1)
    np.savetxt(path, rows, delimiter=",", header=",".join(cols), comments="", fmt="%.9g")
    return path


class ClinicalCsvDataset(torch.utils.data.Dataset):
    """(features, events, durations) per patient from a csv written by `write_synthetic_csv` / shaped like it."""

    def __init__(self, path, predictors):
        with open(path) as f:
            header = f.readline().strip().split(",")
        table = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
        col = {name: i for i, name in enumerate(header)}
        missing = [p for p in predictors if p not in col]
        if missing:
            raise ValueError(f"csv {path} lacks predictor columns {missing[:4]}...")
        self.clinical = torch.from_numpy(table[:, [col[p] for p in predictors]]).float()
        self.events = torch.from_numpy(table[:, [col[f"event{i}"] for i in range(NUM_CLASSES)]]).long()
        self.durations = torch.from_numpy(table[:, [col[f"duration{i}"] for i in range(NUM_CLASSES)]]).long()

    def __len__(self):
        return self.clinical.shape[0]

    def __getitem__(self, i):
        return self.clinical[i], self.events[i], self.durations[i]


def collate(batch):
    xs, ev, du = zip(*batch)
    if isinstance(xs[0], dict):
        x = {k: torch.stack([b[k] for b in xs]).float() for k in xs[0]}
    else:
        x = torch.stack(xs).float()
    return x, torch.stack(ev), torch.stack(du)


def to_device(x, device):
    return {k: v.to(device) for k, v in x.items()} if isinstance(x, dict) else x.to(device)


# ---- survival (main.py:385-601) ----------------------------------------------------------------------------------------------
def train_survival(model, train_ds, val_ds, args, device, rank, world):
    loader = torch.utils.data.DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, collate_fn=collate, drop_last=len(train_ds) > args.batch_size)
    val_loader = torch.utils.data.DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, collate_fn=collate, drop_last=len(val_ds) > args.batch_size)
    model = model.to(device)
    D.broadcast_parameters(model)
    for m in model.modules():                 # this loop only changes weights through FusedSGD: repack them once per optimizer step,
        if hasattr(m, "repack_policy"):       # not on each of the 64 / batch micro-batch forwards in between
            m.repack_policy = "versioned"
    interval = super_batch_interval(args.batch_size, world)
    steps_per_epoch = optimizer_steps_per_epoch(len(loader), interval)
    opt = FusedSGD(model, lr=args.lr, momentum=args.momentum, nesterov=True, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, steps_per_epoch=steps_per_epoch, epochs=args.epochs)
    blender = GradientBlender(CoxPH, survival=True, surv_criterion=surv_criterion) if args.blend else None
    # gradients are reduced once per accumulation window: the window's LAST backward is armed, so the all-reduce of a dense block's
    # gradients starts inside that backward as soon as they are final and overlaps the kernels of the blocks still to come
    reducer = D.OverlappedGradientReducer(model) if world > 1 else None
    cat_dim = 1 if args.blend else 0
    best = float("inf")
    for epoch in range(args.epochs):
        model.train()
        losses, c_pred, c_ev, c_du = [], [], [], []
        for i, (x, ev, du) in enumerate(loader):
            x, ev, du = to_device(x, device), ev.to(device), du.to(device)
            out = model(x)
            loss = blender.computeLoss(out, ev, du)[0] if args.blend else surv_criterion(CoxPH, out, ev, du, device)
            boundary = is_step_boundary(i, len(loader), interval)
            if boundary and reducer is not None:
                reducer.arm()
            loss.backward()
            losses.append(loss.detach())
            if boundary:
                if reducer is not None:
                    reducer.finish()
                opt.step()
                sched.step()
                opt.zero_grad()
            c_pred.append(out.detach()); c_ev.append(ev); c_du.append(du)
        cp = torch.cat(c_pred, dim=cat_dim)
        ce, cd = torch.cat(c_ev), torch.cat(c_du)
        tr_c = getCIndices((cp[0] if args.blend else cp).cpu().numpy(), ce.cpu().numpy(), cd.cpu().numpy())
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
        yp = torch.cat(y_pred, dim=cat_dim)
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
                logger.info('saved new best metric model')
        if args.blend and blender_update_due(epoch, args.blend_update_interval):
            # every rank evaluates the same epoch-level losses: its own training patients are gathered from all ranks (the
            # validation set is identical on every rank), so one global weight vector results, as on a single GPU
            blender.updateWeights(gather_rows(cp, 1, world), gather_rows(ce, 0, world), gather_rows(cd, 0, world), yp, ye, yd)
            if rank == 0:
                logger.info('Completed updating gradient blender weights - new weights : {}'.format(blender.weights))
    if args.blend and rank == 0:
        blender.saveHistory()
    return model


# ---- classification (main.py:125-327) ----------------------------------------------------------------------------------------
def train_classification(model, train_ds, val_ds, args, device, rank=0, world=1):
    """Binary classification of the event flags: pos-weighted BCE on logits (:147-153), SGD-Nesterov + OneCycleLR stepped every
    batch (:207-215), F1 per class (:217-233,:289-300), best mean validation F1 -> model.pth (:301-306); with --blend the
    GradientBlender's classification branch (:156,:210,:264,:311-314).  With `world` > 1 every rank trains on its own patients:
    identical initial weights, gradients SUM-all-reduced before every optimizer step, checkpoints written by rank 0 only."""
    bs = max(2, args.batch_size)
    loader = torch.utils.data.DataLoader(train_ds, batch_size=bs, shuffle=True, collate_fn=collate, drop_last=len(train_ds) > bs)
    val_loader = torch.utils.data.DataLoader(val_ds, batch_size=bs, shuffle=False, collate_fn=collate)
    model = model.to(device)
    D.broadcast_parameters(model)
    freqs = torch.tensor(args.class_frequencies, dtype=torch.float32)
    pos_weights = ((torch.ones_like(freqs) - freqs) / freqs).to(device)
    train_loss_function = BCEWithLogitsLoss(pos_weight=pos_weights, reduction='sum')
    loss_function = BCEWithLogitsLoss(pos_weight=pos_weights, reduction='none')
    opt = torch.optim.SGD(model.parameters(), args.lr, momentum=args.momentum, nesterov=True, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr, steps_per_epoch=len(loader), epochs=args.epochs)
    blender = GradientBlender(loss_function, device=device) if args.blend else None
    best_metric, best_epoch = -1.0, -1
    for epoch in range(args.epochs):
        model.train()
        epoch_loss = torch.zeros((), device=device)
        tps = fps = fns = 0
        train_preds, train_gt, val_preds, val_gt = [], [], [], []
        for x, labels, _ in loader:
            x, labels = to_device(x, device), labels.to(device)
            opt.zero_grad()
            outputs = model(x)
            loss = blender.computeLoss(outputs, labels.float()) if args.blend else criterion(train_loss_function, outputs, labels.float(), device)
            loss.backward()
            D.allreduce_gradients(model)
            opt.step()
            sched.step()
            epoch_loss += loss.detach()
            probs = torch.sigmoid(outputs.detach())
            if args.blend:
                train_preds.append(probs); train_gt.append(labels)
                probs = probs[0]
            hit = probs > CLASSIFICATION_THRESHOLD
            tps = tps + (hit & (labels == 1)).sum(0)
            fps = fps + (hit & (labels == 0)).sum(0)
            fns = fns + (~hit & (labels == 1)).sum(0)
        train_f1 = float(np.mean(getF1Score(tps, fps, fns)))
        model.eval()
        test_loss = 0.0
        y_pred, y = [], []
        with torch.no_grad():
            for x, labels, _ in val_loader:
                x, labels = to_device(x, device), labels.to(device)
                p = model(x)
                l = blender.computeLoss(p, labels.float(), no_reduce=True) if args.blend else criterion(loss_function, p, labels.float(), device)
                test_loss += float(l.sum())
                hit = torch.sigmoid(p) > CLASSIFICATION_THRESHOLD
                if args.blend:
                    val_preds.append(hit.float()); val_gt.append(labels)       # thresholded, as upstream :268-272 feeds the blender
                    hit = hit[0]
                y_pred.append(hit); y.append(labels)
        yp, yt = torch.cat(y_pred), torch.cat(y)
        f1s = getF1Score(((yp == 1) & (yt == 1)).sum(0), ((yp == 1) & (yt == 0)).sum(0), ((yp == 0) & (yt == 1)).sum(0))
        mean_f1 = float(np.nanmean(f1s)) if not np.all(np.isnan(f1s)) else 0.0
        if mean_f1 > best_metric:
            best_metric, best_epoch = mean_f1, epoch + 1
            if rank == 0:
                os.makedirs(args.output_path, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(args.output_path, 'model.pth'))
                logger.info('saved new best metric model')
        if rank == 0:
            logger.info(f"epoch {epoch + 1}/{args.epochs} average loss: {float(epoch_loss) / len(train_ds):.4f} train f1 {train_f1:.4f} "
                        f"validation loss {test_loss / len(val_ds):.4f} current f1: {mean_f1:.4f} best f1: {best_metric:.4f} at epoch: {best_epoch}")
        if args.blend and blender_update_due(epoch, args.blend_update_interval):
            # as in the survival loop: the training patients of all ranks (the validation set is the same everywhere) -> one weight vector
            blender.updateWeights(gather_rows(torch.cat(train_preds, dim=1), 1, world), gather_rows(torch.cat(train_gt).float(), 0, world),
                                  torch.cat(val_preds, dim=1), torch.cat(val_gt).float())
            if rank == 0:
                logger.info('Completed updating gradient blender weights - new weights : {}'.format(blender.weights))
    if rank == 0:
        torch.save(model.state_dict(), os.path.join(args.output_path, 'final_model.pth'))
    return model


def inference_survival(model, ds, args, device):
    """main.py:750-887: batch-1 loop, Grad-CAM maps (saved as .npy; NIfTI export needs nibabel, host I/O), C-index."""
    model = model.to(device).eval()
    if args.bootstrap:
        args.no_gradcam = True                       # main.py:774-777: no attention maps, no prediction dump while bootstrapping
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
    if args.bootstrap:
        means, stds, used = bootstrap_c_indices(p, e, d)
        logger.info('Mean c indices: {}'.format(means))
        logger.info('Std. devs: {} ({} of {} resamples had admissible pairs)'.format(stds, used, NUM_BOOTSTRAP_ITERATIONS))
    else:
        logger.info('All C-indexes: {}'.format(getCIndices(p, e, d)))
    return p


def build_arg_parser():
    ap = argparse.ArgumentParser()
    for flag, h in (("preop", "clinical features available pre-operation"), ("postop", "pre + post operation clinical features"),
                    ("radiomics", "radiomic features (not implemented upstream either)"), ("images", "image data"),
                    ("classification", "binary classification"), ("survival", "time-to-event model"), ("segmentation", "unsupported"),
                    ("lr_finder", "unsupported tooling"), ("no_gradcam", "disable Grad-CAM for inference"), ("inference", "inference"),
                    ("split", "create a new dataset split"), ("blend", "gradient blending"), ("bootstrap", "bootstrap evaluation (with --inference --survival)")):
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
    return ap


def main(argv=None):
    a = build_arg_parser().parse_args(argv)
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
    if a.bootstrap and not (a.inference and a.survival):
        raise SystemExit("--bootstrap resamples the evaluation of `--inference --survival` (main.py:767-887); it has no meaning for training runs")
    if a.image_loc:
        raise SystemExit("image loaders (NIfTI / DICOM / S3) are host I/O outside this path; run without --image_loc for synthetic volumes")

    parser = Parser(a.config)
    cfg = parser.parseConfig()
    hp = cfg.get("Hyperparameters", {})
    a.multimodal = a.images and (a.preop or a.postop)
    a.blend = a.blend and a.multimodal
    a.batch_size = int(hp.get("train_batch_size", 2)) if a.config else 2
    a.momentum, a.weight_decay = float(hp.get("momentum", 0.9)), float(hp.get("weight_decay", 1e-4))
    a.class_frequencies = list(hp.get("class_frequencies", [0.4] * NUM_CLASSES))     # CLASS_FREQUENCIES is undefined upstream (Q1)
    torch.manual_seed(int(hp.get("seed", 42)))
    model = parser.getModel(a)
    if a.multimodal:
        model.blend = a.blend
    rank, world, local = D.init_from_env(os.environ.get("MMNN_DIST_BACKEND", "nccl"))    # "nccl" = RCCL; gloo only for rehearsals on one card
    if not torch.cuda.is_available():
        raise SystemExit("mmnn_sts_amd runs on the MI355X only (no CPU path)")
    local = local % max(1, torch.cuda.device_count())            # (gloo rehearsals: several ranks share a card)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if a.weights:
        model = loadWeights(model, a.weights, "cpu")
    predictors = parser.predictors(a)
    inch = cfg["ImageModel"]["in_channels"]
    mk = lambda n, seed: SyntheticPatients(n, inch, a.synthetic_size, len(predictors), a.multimodal, a.images, seed)
    if not a.images:
        # tabular-only: patients travel through a csv, as upstream's clinical datasets do
        os.makedirs(a.output_path, exist_ok=True)
        n_train, n_val = max(4, a.synthetic_patients), max(4, a.synthetic_patients // 4)
        train_csv = a.data_loc or write_synthetic_csv(os.path.join(a.output_path, f"synthetic_train_rank{rank}.csv"), n_train, predictors, 1000 + rank)
        val_csv = write_synthetic_csv(os.path.join(a.output_path, f"synthetic_val_rank{rank}.csv"), n_val, predictors, 7)
        eval_csv = a.data_loc or val_csv        # --inference --data_loc x.csv evaluates THAT file
        mk = lambda n, seed: ClinicalCsvDataset(train_csv if seed >= 1000 else (eval_csv if seed == 99 else val_csv), predictors)
    elif a.data_loc:
        raise SystemExit("clinical csv + image loaders are host I/O outside this path; run without --data_loc for synthetic patients")
    if a.inference:
        inference_survival(model, mk(max(2, a.synthetic_patients // 4), 99), a, device)
    elif a.survival:
        train_survival(model, mk(a.synthetic_patients, 1000 + rank), mk(max(2, a.synthetic_patients // 4), 7), a, device, rank, world)
    else:
        train_classification(model, mk(max(4, a.synthetic_patients), 1000 + rank), mk(max(4, a.synthetic_patients // 4), 7), a, device, rank, world)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
