"""Cross-rank semantics of the DDP training path, world_size 2 over gloo on the CPU (the GPU node runs the same host
logic over RCCL):
  * every rank runs the same number of steps per epoch and leaves the epoch loop together (ranks with
    different step counts pair gradient all-reduces of different epochs and finally hang);
  * the blurred target is divided by the maximum over the WHOLE batch (reference main.py:230), so shards MAX-all-reduce
    their local maxima: mean of the shard gradients == single-process gradient even when only one shard has
    overlapping echoes (a larger local maximum)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import train_oracle as to
from stofnet_amd import synth
from stofnet_amd.sharding import agree_any, rank_batches
from stofnet_amd.training import allreduce_max_, allreduce_mean_


def test_rank_batches_same_step_count_on_every_rank():
    for n in [0, 1, 7, 15, 16, 1000]:
        for w in [1, 2, 3, 8]:
            per_rank = [rank_batches(n, r, w) for r in range(w)]
            assert len({len(b) for b in per_rank}) == 1                       # equal step counts
            flat = sorted(b for br in per_rank for b in br)
            assert flat == list(range((n // w) * w))                          # a partition of the kept batches
    with pytest.raises(ValueError):
        rank_batches(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_loop_worker(rank, world, port, n_batches, epochs, stop_epoch_rank1):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        steps, done_epochs = 0, 0
        for e in range(epochs):
            for b in rank_batches(n_batches, rank, world):                     # main.train's batch loop
                g = torch.full((4,), float(b + 100 * e))
                allreduce_mean_(g)                                             # one gradient all-reduce per step
                other = [x for x in range(world) if x != rank]
                k = (b - rank) // world                                        # same step index on every rank
                expect = np.mean([r + k * world + 100 * e for r in range(world)])
                assert abs(float(g[0]) - expect) < 1e-9, 'an all-reduce paired steps of different epochs'
                steps += 1
            done_epochs += 1
            want_stop = (rank == 1 and e == stop_epoch_rank1)                  # only one rank's criterion fires
            if agree_any(want_stop):
                break
        out = torch.tensor([steps, done_epochs])
        gathered = [torch.zeros_like(out) for _ in range(world)]
        dist.all_gather(gathered, out)
        assert all(torch.equal(gathered[0], t) for t in gathered)             # same steps, same epochs everywhere
        assert int(out[1]) == stop_epoch_rank1 + 1 and int(out[0]) == (n_batches // world) * (stop_epoch_rank1 + 1)
    finally:
        dist.destroy_process_group()


def test_non_divisible_batch_count_keeps_ranks_in_step():
    """config.yaml's own case: 64 waveforms at batch 4 leave 15 training batches for 2 ranks."""
    mp.spawn(_train_loop_worker, args=(2, _free_port(), 15, 5, 2), nprocs=2, join=True)


def _target_max_worker(rank, world, port, sd, x, gt, r, full_loss, full_grads):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n = x.shape[0] // world
        xs, gts = x[rank * n:(rank + 1) * n], gt[rank * n:(rank + 1) * n]
        like = torch.zeros((n, 1, x.shape[-1] * r), dtype=torch.float64)
        local_max = to.blurred_target(like, torch.as_tensor(gts)).max().reshape(1)
        gmax = allreduce_max_(local_max.clone())                               # the product's helper (training.py)
        loss, grads, _ = to.loss_and_grads(sd, xs, gts, r, 80, blur_max=float(gmax))
        flat = torch.cat([torch.from_numpy(g).reshape(-1) for g in grads.values()])
        allreduce_mean_(flat)                                                  # DDP: mean of the shard gradients
        ref = torch.cat([torch.from_numpy(g).reshape(-1) for g in full_grads.values()])
        assert float((flat - ref).abs().max()) < 1e-12 * max(1.0, float(ref.abs().max()))
        lt = torch.tensor([loss], dtype=torch.float64)
        allreduce_mean_(lt)
        assert abs(float(lt) - full_loss) < 1e-12
        # and the local maximum really differs between the shards (otherwise the test shows nothing)
        maxes = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(maxes, local_max)
        assert abs(float(maxes[0]) - float(maxes[1])) > 0.05
    finally:
        dist.destroy_process_group()


def test_blurred_target_uses_the_batch_global_maximum():
    r, L = 4, 160
    sd = synth.synth_state_dict(r, seed=11)
    x = synth.synth_echo(4, L, seed=3)
    # rank 0's rows hold two echoes 2 samples apart (their blurs overlap: maximum > the single-echo peak); rank 1's do not
    gt = np.array([[[200, 202]], [[300, 302]], [[100, 400]], [[50, 500]]], np.int64)
    full_loss, full_grads, _ = to.loss_and_grads(sd, x, gt, r, 80)
    mp.spawn(_target_max_worker, args=(2, _free_port(), sd, x, gt, r, full_loss, full_grads), nprocs=2, join=True)


def _moment_worker(rank, world, port):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from stofnet_amd.gradpeak import _moment_reduce
        local = torch.tensor([1.0 + rank, 2.0 + rank, 100.0], dtype=torch.float64)
        # default: local moments like the reference (models/gradpeak.py:18 has no collective); only rank 0 calls --
        # with a hidden all-reduce this would deadlock, and unsharded rows on every rank would count W times
        if rank == 0:
            assert torch.equal(_moment_reduce(local.clone(), None), local)
        dist.barrier()
        # opt-in (sharded=True -> default group; or an explicit group): sum, sum of squares and count add up
        got = _moment_reduce(local.clone(), True)
        assert torch.equal(got, torch.tensor([3.0, 5.0, 200.0], dtype=torch.float64))
        got = _moment_reduce(local.clone(), dist.group.WORLD)
        assert torch.equal(got, torch.tensor([3.0, 5.0, 200.0], dtype=torch.float64))
    finally:
        dist.destroy_process_group()


def test_gradpeak_moment_reduction_is_opt_in():
    mp.spawn(_moment_worker, args=(2, _free_port()), nprocs=2, join=True)
