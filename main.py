#!/usr/bin/env python3
"""Inference entry point with the reference's `python main.py key=value ...` interface
(reference main.py:29-34,133-177,292-347), running the MI355X-native hot path.

What it keeps: config.yaml + CLI merge, seeding (main.py:39-41), the model switch
(stofnet / gradpeak, main.py:133-167), checkpoint lookup by file-name prefix with strict
load_state_dict (main.py:173-177), the eval loop's `model(frame)` -> `mask2coords` ->
`toa_rmse` sequence (main.py:314,320,347), and with `evaluate=False` the training loop
(main.py:199-289: Gaussian-mask loss, AdamW, CosineAnnealingLR per epoch, EarlyStopping on the
summed validation loss, checkpoint `<run_name>_rf-scale<rf>_epoch_<e>.pth`, main.py:423-426) on the
HIP training kernels; launched under torch.distributed.run it becomes DDP (batch sharded over ranks,
one flat gradient all-reduce per step over RCCL).  What it drops: datasets (absent from the reference
mount) and wandb.  Inputs come from `input_file` (.npy) or from synthetic echoes with known onsets.
"""
import json
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

script_path = Path(__file__).parent.resolve()
sys.path.insert(0, str(script_path))

from stofnet_amd import GradPeak, StofNet, mask2coords          # noqa: E402
from stofnet_amd import config as config_mod                     # noqa: E402
from stofnet_amd.metrics import toa_rmse                         # noqa: E402


def load_frames(cfg):
    if cfg.input_file:
        arr = np.load(cfg.input_file).astype(np.float32)
        arr = arr[:, None, :] if arr.ndim == 2 else arr
        return arr / np.abs(arr).max(axis=-1, keepdims=True), None     # NormalizeVol (utils/transforms.py:13)
    from stofnet_amd.synth import synth_echo          # deterministic demo inputs only
    x, onsets = synth_echo(int(cfg.num_waveforms), int(cfg.num_samples), seed=int(cfg.seed), return_onsets=True)
    return x, onsets.astype(np.float32)[:, None]


def main(argv=None):
    cfg = config_mod.merge(config_mod.load(str(script_path / 'config.yaml')), config_mod.from_cli(argv))
    torch.manual_seed(cfg.seed)
    random.seed(cfg.seed)
    np.random.seed(cfg.seed)

    name = str(cfg.model).lower()
    if name == 'stofnet':
        model = StofNet(upsample_factor=cfg.upsample_factor, precision=cfg.precision)
    elif name == 'gradpeak':
        chirp = 'chirp' in str(cfg.data_dir).lower()
        model = GradPeak(threshold=cfg.th, rescale_factor=cfg.rf_scale_factor,
                         echo_max=1 if chirp else float('inf'), onset_opt=chirp)
        cfg.evaluate = True
    else:
        raise Exception('Model not recognized')
    if 'LOCAL_RANK' in os.environ and str(cfg.device) == 'cuda':          # one process per GPU under torch.distributed.run
        cfg.device = f"cuda:{os.environ['LOCAL_RANK']}"
        torch.cuda.set_device(cfg.device)
    model = model.to(cfg.device)
    model.eval()

    if name != 'gradpeak' and cfg.model_file:
        ckpt_dir = Path(cfg.ckpt_dir) if os.path.isabs(str(cfg.ckpt_dir)) else script_path / cfg.ckpt_dir
        prefix = str(cfg.model_file).split('_')[0]
        paths = [fn for fn in sorted(ckpt_dir.iterdir()) if fn.name.startswith(prefix)] if ckpt_dir.is_dir() else []
        if paths:
            model.load_state_dict(torch.load(str(paths[0]), map_location=cfg.device, weights_only=True))

    frames, gt = load_frames(cfg)
    history = train(model, frames, gt, cfg) if (not cfg.evaluate and name == 'stofnet') else None
    es_all, summary = evaluate(model, name, frames, gt, cfg)
    if history is not None:
        summary['train_history'] = history
    if int(os.environ.get('RANK', '0')) == 0:
        print(json.dumps(summary))
    return es_all, summary


def train(model, frames, gt, cfg):
    """main.py:199-289 + 403-410 + 423-426 on the HIP training kernels (stofnet_amd/training.py)."""
    import torch.distributed as dist
    from stofnet_amd.sharding import agree_any, rank_batches
    from stofnet_amd.training import StofNetTrainer
    if gt is None:
        raise RuntimeError('training needs ground-truth onsets (synthetic echoes or a labelled input)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group('nccl', device_id=torch.device(cfg.device))
    tr = StofNetTrainer(model, lr=cfg.lr, weight_decay=cfg.weight_decay, lambda_value=cfg.lambda_value,
                        mask_amplitude=cfg.mask_amplitude, kernel_size=cfg.kernel_size, sigma=cfg.sigma,
                        precision=cfg.train_precision)
    r, bs = int(cfg.upsample_factor), int(cfg.batch_size)
    n_val = max(bs, int(frames.shape[0] * 0.1) // bs * bs)              # held-out tail for early stopping
    tr_x, tr_gt = frames[:-n_val], gt[:-n_val]
    va_x, va_gt = frames[-n_val:], gt[-n_val:]
    mine = rank_batches(tr_x.shape[0] // bs, rank, world)                # the same number of steps on every rank
    best, bad, history = float('inf'), 0, []

    def gt_true_of(g):
        g = torch.from_numpy(np.nan_to_num(g, nan=0.0)).to(cfg.device)
        g = torch.where(g <= 0, torch.zeros_like(g), g)                  # main.py:217
        return torch.round(g.unsqueeze(1) * r).long()                    # main.py:218

    for e in range(int(cfg.epochs)):
        tr.set_lr_cosine(e, int(cfg.epochs), float(cfg.lr))              # CosineAnnealingLR stepped per epoch
        model.train()
        tot = 0.0
        for b in mine:
            sl = slice(b * bs, (b + 1) * bs)
            loss, _ = tr.train_step(torch.from_numpy(tr_x[sl]).to(cfg.device), gt_true_of(tr_gt[sl]))
            tot += float(loss)
        model.eval()
        val = 0.0
        with torch.no_grad():
            for b0 in range(0, va_x.shape[0] - bs + 1, bs):
                pred = model(torch.from_numpy(va_x[b0:b0 + bs]).to(cfg.device))
                val += float(tr.loss(pred, gt_true_of(va_gt[b0:b0 + bs])))
        history.append({'epoch': e, 'lr': tr.lr, 'train_loss': tot / max(len(mine), 1), 'val_loss': val})
        if rank == 0:
            print(json.dumps(history[-1]))
        if val < best - float(cfg.delta):                                # EarlyStopping (utils/early_stop.py)
            best, bad = val, 0
        else:
            bad += 1
        if agree_any(bad >= int(cfg.patience), device=cfg.device):       # every rank leaves at the same epoch
            break
    if rank == 0 and cfg.ckpt_dir:
        ckpt_dir = Path(cfg.ckpt_dir) if os.path.isabs(str(cfg.ckpt_dir)) else script_path / cfg.ckpt_dir
        ckpt_dir.mkdir(exist_ok=True)
        path = ckpt_dir / f"{cfg.run_name}_rf-scale{cfg.rf_scale_factor}_epoch_{e + 1}.pth"
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)
        print(json.dumps({'saved': str(path)}))
    return history


def evaluate(model, name, frames, gt, cfg):
    bs = int(cfg.batch_size)
    results, times = [], []
    with torch.no_grad():
        for b0 in range(0, frames.shape[0] - bs + 1, bs):          # drop_last=True (main.py:111)
            frame = torch.from_numpy(frames[b0:b0 + bs]).to(cfg.device)
            torch.cuda.synchronize()
            tic = time.perf_counter()
            out = model(frame)
            if name == 'stofnet':
                es = mask2coords(out, window_size=cfg.nms_win_size, threshold=cfg.th,
                                 upsample_factor=cfg.upsample_factor)
            else:
                es = out
            torch.cuda.synchronize()
            times.append((time.perf_counter() - tic) / bs)
            results.append(es.reshape(bs, -1).cpu().numpy())
    kmax = max(r.shape[1] for r in results)
    es_all = np.concatenate([np.pad(r, ((0, 0), (0, kmax - r.shape[1]))) for r in results], 0)
    summary = {'model': name, 'waveforms': int(es_all.shape[0]), 'inference_time': float(np.mean(times)),
               'waveforms_per_s': float(1.0 / np.mean(times))}
    if gt is not None:
        # toa_rmse (main.py:347) on the device kernel
        errs = toa_rmse(torch.from_numpy(gt[:es_all.shape[0]]).to(cfg.device), torch.from_numpy(es_all).to(cfg.device),
                        tol=cfg.etol).cpu()
        summary['total_distance_mean'] = float(np.nanmean(errs[:, 0].numpy()))
        summary['total_jaccard'] = float(np.nanmean(errs[:, 3].numpy()))
    return es_all, summary


if __name__ == '__main__':
    main()
