#!/usr/bin/env python3
"""bench.py -- StofNet inference throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (StofNet.forward: SemiGlobalBlock kernels + fused body
sweep incl. the SampleShuffle1D store) over one resident batch of synthetic waveforms:
configuration C2 of BASELINE.json, fp32 [4096,1,2000] -> [4096,1,20000] at upsample_factor
10 with seeded-random weights (all shipped checkpoints have r=4).
Rows are independent, so N GPUs each own a [4096,1,2000] batch (weak scaling) and the forward
has no collective; the optional gather of onset indices over RCCL is timed separately.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
dominant kernel (body sweep, MFMA-bound, timed with HIP events on its launch stream) and
`cpu_baseline` (the CPU oracle = PyTorch-CPU restatement, timed on this box's host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, L, R = 4096, 2000, 10
PREHEAT_STEPS = 25      # untimed forward passes before the W warm-up steps (≈0.5 s of GPU time)
PEAK_TFLOPS = {'fp32': 157.3, 'f16x3': 2500.0 / 3.0}   # MI355X_MICROARCH.md: fp32 MFMA 157.3; f16 2.5 PF / 3 passes


def body_flops(n, l, r):
    """conv1 + conv2..12 + conv_last, the work of the body-sweep kernel."""
    return 2.0 * n * l * (576 + 11 * 28672 + 192 * r)


def total_flops(n, l, r):
    return 2.0 * n * (l * (576 + 163840 + 11 * 28672 + 192 * r) + (l // 80) * 163840)


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (a GPU box exposes
    256 hardware threads but grants one job far fewer)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    cores = min(cores, max(1, q // per))
            break
        except Exception:  # noqa: BLE001
            continue
    env = os.environ.get('STOF_CPU_THREADS')
    if env:
        cores = int(env)
    return min(cores, 64)


def cpu_baseline(sd, r, y_gpu=None, idx_gpu=None, sample_rows=256, reps=3):
    """The oracle (oracle/stofnet_oracle.py, PyTorch CPU fp32) on a bounded sample: its speed on the host cores, and
    -- as the checker -- the parity of the timed GPU outputs on the same rows (the input of rank 0 is the same seed)."""
    from oracle import pickers_oracle as po
    from oracle import stofnet_oracle as so
    from stofnet_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    x = synth.synth_randn(sample_rows, L, seed=3008)
    with torch.no_grad():
        so.stofnet_forward(sd, x[:16], r)          # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            y_ref = so.stofnet_forward(sd, x, r)
        dt = (time.perf_counter() - t0) / reps
    out = {'value': round(sample_rows / dt, 2), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
           'sample': f'[{sample_rows},1,{L}] fp32, upsample_factor={r}, PyTorch-CPU oracle, '
                     f'{reps} reps after warm-up, torch {torch.__version__}'}
    if y_gpu is not None:
        y_ref = y_ref.numpy()
        ref_idx = po.maxima_positions(y_ref, 20, None)       # int64 [K, 2] (row, index), arg-max mode
        first = np.full(sample_rows, -1, np.int64)
        for row, t in ref_idx[::-1]:
            first[row] = t
        got = (idx_gpu[:sample_rows, 0].cpu().numpy().astype(np.int64) if idx_gpu.shape[1] > 0
               else np.full(sample_rows, -1, np.int64))
        out['parity_on_sample'] = {
            'onset_index_mae': float(np.abs(got - first).mean()),
            'onset_index_mismatches': int((got != first).sum()),
            'max_rel_err_maps': float(np.abs(y_gpu[:sample_rows].cpu().numpy() - y_ref).max() / np.abs(y_ref).max())}
    return out


def train_bench(args, dev, dist, rank, world):
    """BASELINE.json configs[4]: one training step = forward (activations kept) + Gaussian-mask loss + backward +
    mean all-reduce of the single 2.58 MB gradient bucket (RCCL, N > 1) + AdamW, exact fp32, batch per GPU fixed."""
    from stofnet_amd import synth                      # deterministic inputs/weights only
    from stofnet_amd import StofNet
    from stofnet_amd.training import StofNetTrainer
    sd = synth.synth_state_dict(R, seed=3008)
    model = StofNet(upsample_factor=R)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    tr = StofNetTrainer(model.to(dev), precision=args.train_precision)
    nb = args.train_batch
    x = torch.from_numpy(synth.synth_echo(nb, L, seed=3008 + rank)).to(dev)
    rng = np.random.default_rng(rank)
    gt = torch.from_numpy(np.sort(rng.integers(1, L * R, size=(nb, 1, 2)), -1)).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        loss, _ = tr.train_step(x, gt)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.train_step(x, gt)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        flops = 3.0 * total_flops(nb, L, R)        # forward + data-gradient + weight-gradient
        achieved = flops * args.steps / dt / 1e12
        out = {
            'metric': 'RF waveforms/sec StofNet training step (fwd+bwd+AdamW) rf_scale=10',
            'value': round(world * nb * args.steps / dt, 1), 'unit': 'waveforms/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.train_precision == 'fp32' else 'f32 via split-fp16 x3 MFMA operands (hi+lo, fp32 accumulate) in fwd, dgrad and wgrad convolutions',
            'data': 'synthetic',
            'config': {'workload': f'C5 training step [{nb},1,{L}] -> [{nb},1,{L * R}] per GPU, Gaussian-mask loss, AdamW, '
                                   f'upsample_factor={R}', 'rows_per_gpu': nb, 'L': L, 'upsample_factor': R,
                       'parallelism': f'ddp{world}: one flat 2.58 MB gradient all-reduce per step'},
            'roofline': {'bound': 'mfma', 'kernel': 'whole step (conv_cl_kernel + conv_wgrad_cl_kernel, fp32 MFMA)',
                         'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS['fp32'], 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_TFLOPS['fp32'], 4), 'traffic': None},
            'final_loss': float(loss),
        }
        if world == 1 and not args.no_cpu_baseline:
            # the oracle's training step (torch autograd on the host cores), bounded sample
            from oracle import train_oracle
            cores = host_cores()
            torch.set_num_threads(cores)
            rows = 32
            xs, gts = x[:rows].cpu().numpy(), gt[:rows].cpu().numpy()
            train_oracle.loss_and_grads(sd, xs[:4], gts[:4], R, 80, dtype=torch.float32)
            t1 = time.perf_counter()
            train_oracle.loss_and_grads(sd, xs, gts, R, 80, dtype=torch.float32)
            cdt = time.perf_counter() - t1
            out['cpu_baseline'] = {'value': round(rows / cdt, 2), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
                                   'sample': f'1 fwd+bwd of {rows} waveforms (oracle, torch autograd fp32), optimizer step excluded'}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default=os.environ.get('STOF_PRECISION', 'f16x3'), choices=['fp32', 'f16x3'])
    ap.add_argument('--no-fp32-extra', action='store_true', help='skip the secondary exact-fp32 measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="'train' times BASELINE.json configs[4] (fwd+bwd+AdamW, DDP gradient all-reduce) instead")
    ap.add_argument('--train-batch', type=int, default=256, help='waveforms per GPU per training step')
    ap.add_argument('--train-precision', default='fp32', choices=['fp32', 'f16x3'],
                    help='arithmetic of the forward / data-gradient convolutions of the training step')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    if args.mode == 'train':
        return train_bench(args, dev, dist, rank, world)

    from stofnet_amd import synth                      # deterministic inputs/weights only (not the oracle math)
    from stofnet_amd import StofNet, _lib
    from stofnet_amd.mask2samples import onset_indices

    sd = synth.synth_state_dict(R, seed=3008)
    model = StofNet(upsample_factor=R, precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    x = torch.from_numpy(synth.synth_randn(N_ROWS, L, seed=3008 + rank)).to(dev)   # resident before timing

    lib = _lib.lib()
    nev = 4
    ev_sets = []
    for _ in range(args.steps):
        arr = (ctypes.c_void_p * nev)()
        _lib.check(lib.stof_events_create(nev, arr))
        ev_sets.append(arr)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(PREHEAT_STEPS):                 # untimed: lets the clocks/power state settle whatever W is
        y = model(x)
    for _ in range(args.warmup):
        y = model(x)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        y = model(x, _events=ev_sets[s])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel durations from the HIP events recorded inside the timed region
    kern = np.zeros((args.steps, 3))
    ms = ctypes.c_float()
    for s, arr in enumerate(ev_sets):
        for k in range(3):
            _lib.check(lib.stof_event_elapsed_ms(arr[k], arr[k + 1], ctypes.byref(ms)))
            kern[s, k] = ms.value
        lib.stof_events_destroy(nev, arr)
    k_ms = kern.mean(0)

    # extras outside the timed region: picker and optional RCCL gather of the onset indices
    counts, idx = onset_indices(y, 20, None)          # warm-up: first use loads the picker's code object
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        counts, idx = onset_indices(y, 20, None)      # includes the Kmax host sync the reference has too
    torch.cuda.synchronize()
    pick_ms = (time.perf_counter() - t1) * 1e3 / 5
    gather_ms = None
    if dist is not None:
        onset = (idx[:, 0] if idx.shape[1] > 0 else torch.zeros(idx.shape[0], dtype=idx.dtype, device=dev)).contiguous()
        outs = [torch.empty_like(onset) for _ in range(world)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.all_gather(outs, onset)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3

    # secondary measurement: the exact-fp32 parity-baseline mode, same workload (outside the timed region)
    fp32_extra = None
    if args.precision != 'fp32' and not args.no_fp32_extra:
        m32 = StofNet(upsample_factor=R, precision='fp32')
        m32.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        m32 = m32.to(dev).eval()
        m32(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            y32 = m32(x)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t1) / 3
        c32, i32 = onset_indices(y32, 20, None)
        fp32_extra = {'waveforms_per_s_per_gpu': round(N_ROWS / dt32, 1), 'ms_per_step': round(dt32 * 1e3, 3),
                      'max_rel_diff_vs_timed_mode': float((y32 - y).abs().max() / y32.abs().max()),
                      'onset_index_mismatches_vs_timed_mode': int((i32[:, :1] != idx[:, :1]).sum())}
        del m32, y32

    if rank == 0:
        value = world * N_ROWS * args.steps / dt
        body_s = k_ms[2] * 1e-3
        achieved = body_flops(N_ROWS, L, R) / body_s / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('body_sweep_hbm_bytes_per_launch')
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            'metric': 'RF waveforms/sec StofNet inference rf_scale=10',
            'value': round(value, 1), 'unit': 'waveforms/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'preheat_steps': PREHEAT_STEPS, 'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else 'f32 via split-fp16 x3 MFMA operands (hi+lo, fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': f'C2 StofNet.forward [{N_ROWS},1,{L}] -> [{N_ROWS},1,{L * R}] per GPU, '
                                   f'upsample_factor={R}, SemiGlobalBlock on, seeded-random weights (seed 3008)',
                       'rows_per_gpu': N_ROWS, 'L': L, 'upsample_factor': R, 'sharding': f'batch x{world}, no collective',
                       'precision': args.precision},
            'roofline': {'bound': 'mfma', 'kernel': 'body_sweep_kernel', 'achieved': round(achieved, 2),
                         'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'flops_per_launch': body_flops(N_ROWS, L, R), 'avg_launch_ms': round(float(k_ms[2]), 4)},
            'kernels_ms': {'sgb_contract_pool': round(float(k_ms[0]), 4), 'sgb_expand': round(float(k_ms[1]), 4),
                           'body_sweep': round(float(k_ms[2]), 4)},
            'whole_forward_tflops': round(total_flops(N_ROWS, L, R) * world * args.steps / dt / 1e12, 2),
            'extras': {'picker_argmax_ms': round(pick_ms, 3),
                       'waveforms_per_s_forward_plus_picker': round(world * N_ROWS / (dt / args.steps + pick_ms * 1e-3), 1),
                       'index_gather_ms': None if gather_ms is None else round(gather_ms, 3),
                       'fp32_exact_mode': fp32_extra},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, R, y, idx)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
