"""Batch sharding of the hot path over the GPUs of one node.

Every row of [N,1,L] is independent in StofNet.forward, the shuffle, the Hilbert envelope and
(row-wise) both pickers, so rank g owns a contiguous block of rows and the forward needs no
collective.  The reference has no distributed code at all; this module is new work and uses
torch.distributed only (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests):

  * shard_rows          -- contiguous row blocks N/G per rank
  * gather_onsets       -- optional result gather: one MAX all-reduce of Kmax (mask2coords pads
                           to the batch-wide Kmax, utils/mask2samples.py:93) + one all_gather of
                           the [rows, Kmax] int32 onset indices (4*K bytes per waveform)
  * global_grad_moments -- GradPeak's default threshold uses the std of the WHOLE batch
                           (models/gradpeak.py:18, Q7): all-reduce (sum, sum of squares, count); gradpeak.py does this
                           on the device tensor when torch.distributed is initialised
  * rank_batches        -- DDP training: every rank runs the SAME number of steps per epoch (each step is one
                           gradient all-reduce; unequal counts would pair collectives of different epochs and hang)
  * agree_any           -- a boolean decision (early stopping) taken identically on every rank
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of the contiguous row block of `rank`; the first n_rows % world ranks get one more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad rank/world')
    base, extra = divmod(n_rows, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_onsets(counts: torch.Tensor, idx: torch.Tensor, group=None):
    """counts[rows] int32, idx[rows, K_local] int32 (this rank's shard) -> (counts_all[N], idx_all[N, Kmax])
    on every rank, rows in rank order, zero padded to the batch-wide Kmax."""
    world = dist.get_world_size(group)
    k_local = torch.tensor([idx.shape[1] if idx.dim() == 2 else 0], dtype=torch.int64, device=counts.device)
    rows = torch.tensor([counts.shape[0]], dtype=torch.int64, device=counts.device)
    dist.all_reduce(k_local, op=dist.ReduceOp.MAX, group=group)
    kmax = int(k_local.item())
    rows_all = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(rows_all, rows, group=group)
    rows_all = [int(r.item()) for r in rows_all]
    rmax = max(rows_all)
    pad = torch.zeros((rmax, kmax + 1), dtype=torch.int32, device=counts.device)
    pad[:counts.shape[0], 0] = counts
    if kmax and idx.numel():
        pad[:idx.shape[0], 1:1 + idx.shape[1]] = idx
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    full = torch.cat([o[:r] for o, r in zip(outs, rows_all)], 0)
    counts_all = full[:, 0].contiguous()
    idx_all = full[:, 1:].contiguous()
    # entries beyond a row's count are padding
    ar = torch.arange(kmax, device=counts.device)[None, :]
    idx_all = torch.where(ar < counts_all[:, None], idx_all, torch.zeros_like(idx_all))
    return counts_all, idx_all


def global_grad_moments(s1: float, s2: float, count: int, device=None, group=None):
    """All-reduce the three moments behind GradPeak's batch-wide default threshold (Q7)."""
    t = torch.tensor([s1, s2, float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]), float(t[1]), int(round(float(t[2])))


def rank_batches(n_batches: int, rank: int, world: int) -> list[int]:
    """Global batch indices of `rank` for one epoch: rank, rank + world, ... over the first
    (n_batches // world) * world batches.  Every rank gets exactly n_batches // world steps -- the remainder
    (< world batches) is dropped, like DataLoader(drop_last=True) drops a ragged tail -- so the per-step gradient
    all-reduces line up across ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad rank/world')
    steps = n_batches // world
    return [rank + k * world for k in range(steps)]


def agree_any(flag: bool, device=None, group=None) -> bool:
    """True on every rank iff `flag` is true on at least one (MAX all-reduce of one byte-sized int)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))
