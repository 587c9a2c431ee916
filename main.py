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

from stofnet_amd import EDSR_1D, ESPCN_1D, GradPeak, StofNet, mask2coords          # noqa: E402
from stofnet_amd import config as config_mod                     # noqa: E402
from stofnet_amd.metrics import toa_rmse                         # noqa: E402


class RunLog:
    """The reference's `logging` switch (main.py:113-130): falsy = no logging at all; otherwise the run is logged under that
    group name -- to wandb when it is installed (same project, metric names and summary keys as main.py:115-130,349-373,
    415-421), else to `<run_name>_<group>.jsonl` next to the script with the same keys, so downstream table scripts find
    the same fields."""

    SUMMARY_KEYS = ('model_name', 'total_parameters', 'total_jaccard', 'total_inference_time', 'total_distance_mean',
                    'total_distance_std')

    def __init__(self, cfg):
        self.enabled = bool(cfg.logging) and int(os.environ.get('RANK', '0')) == 0
        self.wb = None
        self.path = None
        self.name = str(cfg.run_name)
        if not self.enabled:
            return
        try:
            import wandb
            self.wb = wandb.init(project='StofNet', resume='allow', anonymous='must', config=cfg.to_container(),
                                 group=str(cfg.logging))
            self.name = self.wb.name
        except ImportError:
            self.path = script_path / f'{cfg.run_name}_{cfg.logging}.jsonl'
            self.path.write_text('')

    def log(self, record):
        if not self.enabled:
            return
        if self.wb is not None:
            self.wb.log(record)
        else:
            with open(self.path, 'a') as f:
                f.write(json.dumps({k: (float(v) if hasattr(v, '__float__') else v) for k, v in record.items()}) + '\n')

    def summary(self, values):
        if not self.enabled:
            return
        if self.wb is not None:
            for k, v in values.items():
                self.wb.summary[k] = v
            self.wb.finish()
        else:
            self.log({'summary': values})


def load_frames(cfg):
    if cfg.input_file:
        arr = np.load(cfg.input_file).astype(np.float32)
        if arr.ndim == 4:                                                # PALA loader output [B, waves, C, S]: main.py:301 takes wave wv_idx = 1 (main.py:71)
            arr = arr[:, int(getattr(cfg, 'wave_index', 1))]
        if arr.ndim == 3 and arr.shape[1] > 1:
            # PALA-shaped frames [B, C, S]: NormalizeVol acts on the whole frame (utils/transforms.py:13), then
            # `frame.reshape(-1, S).unsqueeze(1)` (main.py:301) -> [B*C, 1, S]; batch_size keeps counting frames
            arr = arr / np.abs(arr).max(axis=(1, 2), keepdims=True)
            cfg.rows_per_frame = int(arr.shape[1])
            return arr.reshape(-1, 1, arr.shape[-1]), None
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
    elif name == 'edsr':                                                  # main.py:139-142: baselines riding on SampleShuffle1D
        model = EDSR_1D(num_channels=1, num_features=64, num_blocks=8, upscale_factor=cfg.upsample_factor)
        cfg.evaluate = True
    elif name == 'espcn':
        model = ESPCN_1D(upscale_factor=cfg.upsample_factor)
        cfg.evaluate = True
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
    log = RunLog(cfg)
    if log.enabled and str(cfg.run_name) == 'local-run':
        cfg.run_name = log.name                                          # the reference names checkpoints after the wandb run
    history = train(model, frames, gt, cfg, log) if (not cfg.evaluate and name == 'stofnet') else None
    es_all, summary = evaluate(model, name, frames, gt, cfg, log)
    log.summary({'model_name': name, 'total_parameters': int(sum(p.numel() for p in model.parameters())),
                 'total_jaccard': summary.get('total_jaccard'), 'total_inference_time': summary['inference_time'],
                 'total_distance_mean': summary.get('total_distance_mean'), 'total_distance_std': summary.get('total_distance_std')})
    if history is not None:
        summary['train_history'] = history
    if int(os.environ.get('RANK', '0')) == 0:
        print(json.dumps(summary))
    return es_all, summary


def train(model, frames, gt, cfg, log=None):
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

    autograd = str(getattr(cfg, 'trainer', 'fused')) == 'autograd'
    if autograd:
        # the reference's own training lines (main.py:179-180,184-188,221-248,288) on the module's autograd boundary:
        # torch loss, torch.optim.AdamW, CosineAnnealingLR; `loss.backward()` runs the stof_train_* kernels
        import torch.nn.functional as F
        from stofnet_amd.mask2samples import coords2mask
        from stofnet_amd.training import allreduce_max_, allreduce_mean_, gaussian_kernel
        model.train_precision = str(cfg.train_precision)
        optimizer = torch.optim.AdamW(model.parameters(), lr=float(cfg.lr), weight_decay=float(cfg.weight_decay))
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, int(cfg.epochs))
        loss_mse, loss_l1 = torch.nn.MSELoss(reduction='mean'), torch.nn.L1Loss(reduction='mean')
        gauss = torch.tensor(gaussian_kernel(int(cfg.kernel_size), cfg.sigma), dtype=torch.float32,
                             device=cfg.device).unsqueeze(0).unsqueeze(0)

        def autograd_step(frame, gt_true):
            masks_pred = model(frame)                                                            # main.py:221
            masks_true = coords2mask(gt_true, masks_pred)                                        # main.py:228
            blur = F.conv1d(masks_true, gauss, padding=int(cfg.kernel_size) // 2)                # main.py:229
            blur = blur / allreduce_max_(blur.max().reshape(1))                                  # main.py:230 (whole batch, all ranks)
            blur = blur * float(cfg.mask_amplitude)                                              # main.py:231
            loss = (loss_mse(masks_pred.squeeze(1), blur.squeeze(1).float()) +
                    loss_l1(masks_pred.squeeze(1), torch.zeros_like(masks_pred.squeeze(1))) * float(cfg.lambda_value))   # main.py:232
            optimizer.zero_grad()                                                                # main.py:246
            loss.backward()                                                                      # main.py:247
            if world > 1:                                                                        # DDP: mean of the shard gradients
                for prm in model.parameters():
                    allreduce_mean_(prm.grad)
            optimizer.step()                                                                     # main.py:248
            return loss.detach(), masks_pred.detach()

    for e in range(int(cfg.epochs)):
        if not autograd:
            tr.set_lr_cosine(e, int(cfg.epochs), float(cfg.lr))          # CosineAnnealingLR stepped per epoch
        model.train()
        tot = 0.0
        for b in mine:
            sl = slice(b * bs, (b + 1) * bs)
            step = autograd_step if autograd else tr.train_step
            loss, _ = step(torch.from_numpy(tr_x[sl]).to(cfg.device), gt_true_of(tr_gt[sl]))
            tot += float(loss)
            if log is not None and log.enabled:
                log.log({'train_step': e * len(mine) + (b - rank) // world + 1, 'train_loss': float(loss)})      # main.py:251-255
        if not autograd:
            tr.raise_if_overflow()           # split-fp16 training: the range guard's sticky word, once per epoch (fp32: never set)
        model.eval()
        val = 0.0
        with torch.no_grad():
            for b0 in range(0, va_x.shape[0] - bs + 1, bs):
                pred = model(torch.from_numpy(va_x[b0:b0 + bs]).to(cfg.device))
                val += float(tr.loss(pred, gt_true_of(va_gt[b0:b0 + bs])))
        lr_now = optimizer.param_groups[0]['lr'] if autograd else tr.lr
        if autograd:
            scheduler.step()                                                                                # main.py:288
        history.append({'epoch': e, 'lr': lr_now, 'train_loss': tot / max(len(mine), 1), 'val_loss': val})
        if rank == 0:
            print(json.dumps(history[-1]))
        if log is not None:
            log.log({'lr': lr_now, 'epoch': e})                                                              # main.py:282-286
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


def evaluate(model, name, frames, gt, cfg, log=None):
    bs = int(cfg.batch_size) * int(getattr(cfg, 'rows_per_frame', 1))      # PALA: a batch of B frames is B*C rows (main.py:301)
    results, times = [], []
    with torch.no_grad():
        for b0 in range(0, frames.shape[0] - bs + 1, bs):          # drop_last=True (main.py:111)
            frame = torch.from_numpy(frames[b0:b0 + bs]).to(cfg.device)
            torch.cuda.synchronize()
            tic = time.perf_counter()
            out = model(frame)
            if name in ('stofnet', 'edsr', 'espcn'):                     # main.py:318-320
                es = mask2coords(out, window_size=cfg.nms_win_size, threshold=cfg.th,
                                 upsample_factor=cfg.upsample_factor)
            else:
                es = out
            torch.cuda.synchronize()
            times.append((time.perf_counter() - tic) / bs)
            if name == 'stofnet' and getattr(model, 'precision', '') == 'f16x3':
                model.raise_if_overflow()                    # the fast mode alone has no fp32 re-run ('auto' does)
            results.append(es.reshape(bs, -1).cpu().numpy())
            if log is not None:
                log.log({'val_step': b0 // bs + 1, 'inference_time': times[-1]})                            # main.py:349-356
    kmax = max(r.shape[1] for r in results)
    es_all = np.concatenate([np.pad(r, ((0, 0), (0, kmax - r.shape[1]))) for r in results], 0)
    summary = {'model': name, 'waveforms': int(es_all.shape[0]), 'inference_time': float(np.mean(times)),
               'waveforms_per_s': float(1.0 / np.mean(times))}
    if gt is not None:
        # toa_rmse (main.py:347) on the device kernel
        errs = toa_rmse(torch.from_numpy(gt[:es_all.shape[0]]).to(cfg.device), torch.from_numpy(es_all).to(cfg.device),
                        tol=cfg.etol).cpu()
        dist_all = errs[:, 0].numpy()
        with np.errstate(all='ignore'):
            summary['total_distance_mean'] = float(np.nanmean(dist_all)) if np.isfinite(dist_all).any() else float('nan')
            summary['total_distance_std'] = float(np.std(dist_all[~np.isnan(dist_all)])) if np.isfinite(dist_all).any() else float('nan')
            summary['total_jaccard'] = float(np.nanmean(errs[:, 3].numpy()))
        if log is not None:
            for k, row in enumerate(errs.numpy()):                                                          # main.py:359-373
                log.log({'val_idx': k, 'val_toa_distance': row[0], 'val_toa_precision': row[1], 'val_toa_recall': row[2],
                         'val_toa_jaccard': row[3], 'val_toa_true_positive': row[4], 'val_toa_false_positive': row[5],
                         'val_toa_false_negative': row[6]})
    return es_all, summary


if __name__ == '__main__':
    main()
