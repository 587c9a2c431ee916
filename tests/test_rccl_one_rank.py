"""RCCL itself, on the hardware a build box has: ONE rank, backend "nccl" (= RCCL on ROCm), on cuda:0.

The multi-GPU path (BASELINE.json north_star: "RCCL over xGMI only for the optional result gather")
is covered on the CPU by the world-size-2 gloo tests; what those cannot show is that the tensors the path hands to the
collective library -- int32 index gathers, the float64 moment triple of GradPeak's default threshold, the 2.58 MB fp32 gradient
bucket, MAX all-reduces -- are accepted by RCCL on device buffers.  A one-rank process group exercises exactly that (communicator
set-up, kernel launches of the collectives, dtypes) before an 8-GPU node ever sees the code.

Every process group lives in a CHILD process started with subprocess (never an exec of a process that has touched the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(**extra):
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', LOCAL_WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env.update(extra)
    return env


CHILD = r'''
import datetime, os, sys
import numpy as np
import torch
import torch.distributed as dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
assert dist.get_backend() == 'nccl'
from stofnet_amd import GradPeak, synth
from stofnet_amd.sharding import agree_any, gather_onsets, global_grad_moments
from stofnet_amd.training import allreduce_max_, allreduce_mean_

# 1. the result gather: one MAX all-reduce of Kmax (int64) + all_gather of rows and of the int32 [rows, 1 + Kmax] block
g = torch.Generator().manual_seed(1)
counts = torch.randint(0, 4, (300,), generator=g, dtype=torch.int32)
idx = torch.randint(1, 20000, (300, 3), generator=g, dtype=torch.int32)
idx = torch.where(torch.arange(3)[None, :] < counts[:, None], idx, torch.zeros_like(idx))
c_all, i_all = gather_onsets(counts.to(dev), idx.to(dev))
assert c_all.dtype == torch.int32 and i_all.dtype == torch.int32
assert torch.equal(c_all.cpu(), counts) and torch.equal(i_all.cpu(), idx)

# 2. DDP: the flat 645,764-float gradient bucket (2.58 MB) through a SUM all-reduce (STOF_FORCE_COLLECTIVES: at one rank too)
bucket = torch.randn(645764, generator=g).to(dev)
ref = bucket.clone()
allreduce_mean_(bucket)
torch.cuda.synchronize()
assert torch.equal(bucket, ref)

# 3. the batch-global maximum of the blurred target (float32 [1], MAX) and the early-stopping vote (int32 MAX)
t = torch.tensor([0.25], dtype=torch.float32, device=dev)
assert float(allreduce_max_(t)) == 0.25
v = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX)
assert int(v) == 1 and agree_any(True, device=dev) is True

# 4. GradPeak's default-threshold moments: a float64 triple on the device, SUM all-reduce; and the module with sharded=True
s1, s2, cnt = global_grad_moments(1.5, 2.5, 7, device=dev)
assert (s1, s2, cnt) == (1.5, 2.5, 7)
stats = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(stats, op=dist.ReduceOp.SUM)
assert stats.cpu().tolist() == [1.0, 2.0, 3.0]
x = torch.from_numpy(synth.synth_echo(64, 2000, seed=5)).to(dev)
local = GradPeak(threshold=None, rescale_factor=10, echo_max=1, onset_opt=True)(x)
shard = GradPeak(threshold=None, rescale_factor=10, echo_max=1, onset_opt=True, sharded=True)(x)
assert torch.equal(local, shard)

dist.barrier()
dist.destroy_process_group()
print('RCCL_ONE_RANK_OK')
'''


@pytest.mark.gpu
def test_one_rank_nccl_group_takes_every_collective_of_the_path():
    pr = subprocess.run([sys.executable, '-c', CHILD], env=_env(STOF_FORCE_COLLECTIVES='1'), cwd=ROOT, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == 0 and b'RCCL_ONE_RANK_OK' in pr.stdout, pr.stderr.decode(errors='replace')[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize('config,extra', [('C2', ['--rows', '512']), ('C5', ['--rows', '8'])])
def test_bench_runs_through_a_one_rank_rccl_group(config, extra):
    """bench.py --gpus 1 with the process group forced: Dist creates the nccl group, the barriers / MAX timing all-reduce /
    census run on RCCL, C2 gathers its onset indices over it, C5 all-reduces the gradient bucket every step."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--config', config,
           '--no-cpu-baseline', '--no-fp32-extra', '--no-extra-configs'] + extra
    pr = subprocess.run(cmd, env=_env(STOF_FORCE_PROCESS_GROUP='1', STOF_FORCE_COLLECTIVES='1'), cwd=ROOT, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=400)
    assert pr.returncode == 0, pr.stderr.decode(errors='replace')[-3000:]
    line = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 1 and rec['ranks']['ranks_seen'] == 1
    assert rec['ranks']['collective_backend'] == 'nccl (RCCL)'
    if config == 'C2':
        assert rec['extras']['index_gather_ms'] is not None and rec['extras']['gathered_rows'] == 512
    else:
        assert rec['final_loss'] == rec['final_loss']           # a number, not NaN
