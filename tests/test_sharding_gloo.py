"""world_size-2 gloo tests of the multi-GPU host logic (row sharding + optional onset gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pickers_oracle as po
from stofnet_amd.sharding import gather_onsets, global_grad_moments, shard_rows


def test_shard_rows_partition():
    for n in [0, 1, 7, 4096, 1048576]:
        for w in [1, 2, 3, 8]:
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scores, th):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n = scores.shape[0]
        b, e = shard_rows(n, rank, world)
        local = scores[b:e]
        # each rank picks on its own shard (oracle picker stands in for the GPU kernel on CPU)
        pos = po.maxima_positions(local, 20, th)
        counts = np.bincount(pos[:, 0], minlength=e - b).astype(np.int32)
        k = int(counts.max()) if counts.size else 0
        idx = np.zeros((e - b, k), np.int32)
        fill = np.zeros(e - b, np.int64)
        for r_, t_ in pos:
            idx[r_, fill[r_]] = t_
            fill[r_] += 1
        c_all, i_all = gather_onsets(torch.from_numpy(counts), torch.from_numpy(idx))
        # the gathered result must equal the single-process picker on the whole batch
        ref = po.mask2coords(scores, 20, th, 1)
        got = i_all.numpy().astype(np.float32)
        if ref.ndim == 3:
            assert i_all.shape[1] == 0 or not got.any()
        else:
            assert got.shape == ref.shape and np.array_equal(got, ref), (got.shape, ref.shape)
        assert c_all.shape[0] == n
        s1, s2, cnt = global_grad_moments(float(local.sum()), float((local.astype(np.float64) ** 2).sum()), local.size)
        assert cnt == scores.size and abs(s1 - float(scores.sum())) < 1e-6 * max(1.0, abs(s1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('th', [None, 0.8])
def test_gather_onsets_world2(th):
    rng = np.random.default_rng(5)
    scores = rng.standard_normal((7, 1, 300)).astype(np.float32)
    scores[3, 0, 40] = scores[3, 0, 200] = 9.0       # a tie -> Kmax differs between shards
    mp.spawn(_worker, args=(2, _free_port(), scores, th), nprocs=2, join=True)


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    from stofnet_amd.training import allreduce_mean_
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        allreduce_mean_(g)
        q.put((rank, g.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_ddp_gradient_bucket_mean_allreduce_world2():
    """DDP semantics of the training step's single flat gradient bucket (stofnet_amd/training.py)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + 7
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    expect = np.arange(1000, dtype=np.float32) * 1.5
    assert np.array_equal(res[0], expect) and np.array_equal(res[1], expect)
