#!/usr/bin/env python3
"""bench.py -- StofNet hot-path throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the hot path over one resident batch of synthetic waveforms per GPU:

  C2 (default)  StofNet.forward, fp32 [4096,1,2000] -> [4096,1,20000], upsample_factor 10, seeded weights
                (BASELINE.json configs[1]: the configuration the metric is quoted on)
  C3            the same at upsample_factor 20 -> [4096,1,40000] (configs[2], stresses the SampleShuffle1D store)
  C4            PALA shape (configs[3]): 131,072 rows of 1536 samples per GPU in 8192-row chunks, upsample_factor 4,
                checkpoint graceful-snow, forward + mask2coords picker in threshold mode th = 0.015
                (bash_scripts/array_pala_params.txt:1), ragged onset indices gathered over RCCL afterwards
  C5            training step (configs[4]): forward + Gaussian-mask loss + backward + gradient all-reduce + AdamW

Rows are independent, so N GPUs each own their own batch (weak scaling) and the forward has no collective; the optional
gather of onset indices (one MAX all-reduce of Kmax + one all_gather of int32 indices) is timed separately.

Launch: with --gpus N > 1 and no RANK in the environment this process only spawns N children (one per GPU, RCCL
rendezvous on 127.0.0.1) and relays rank 0's line -- it never touches the GPU itself; under torch.distributed.run the
ranks come from the environment and --gpus must agree with WORLD_SIZE.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel (body sweep,
MFMA-bound, timed with HIP events on its launch stream inside the timed region) and `cpu_baseline` (the CPU oracle =
PyTorch-CPU restatement, timed on this box's host cores, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L_CHIRP = 2000
PREHEAT_SECONDS = 1.5      # untimed passes before the W warm-up steps so the clocks / power state have settled
PEAK_TFLOPS = {'fp32': 157.3, 'f16x3': 2500.0 / 3.0, 'auto': 2500.0 / 3.0}   # MI355X_MICROARCH.md: fp32 MFMA 157.3; f16 2.5 PF / 3 passes
DTYPE_TEXT = {'fp32': 'f32', 'f16x3': 'f32 via split-fp16 x3 MFMA operands (hi+lo, fp32 accumulate)',
              'auto': 'f32 via split-fp16 x3 MFMA operands (hi+lo, fp32 accumulate), device-side exact-f32 re-run on fp16 range overflow'}
C4_ROWS, C4_CHUNK, C4_L, C4_R, C4_TH = 131072, 8192, 1536, 4, 0.015


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C2', choices=['C2', 'C3', 'C4', 'C5'])
    ap.add_argument('--precision', default=os.environ.get('STOF_PRECISION', 'auto'), choices=['auto', 'fp32', 'f16x3'])
    ap.add_argument('--no-fp32-extra', action='store_true', help='skip the secondary exact-fp32 measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'], help="'train' = --config C5")
    ap.add_argument('--train-batch', type=int, default=256, help='waveforms per GPU per training step')
    ap.add_argument('--train-precision', default='f16x3', choices=['fp32', 'f16x3'],
                    help='arithmetic of the convolutions of the training step')
    ap.add_argument('--trainer', default='fused', choices=['fused', 'autograd'],
                    help="C5: 'fused' = StofNetTrainer (loss + AdamW kernels); 'autograd' = the reference's torch loss / "
                         "torch.optim.AdamW lines on the module's autograd boundary (main.py:221-248)")
    ap.add_argument('--rows', type=int, default=0, help='override the rows per GPU of the chosen config (smoke runs)')
    ap.add_argument('--cpu-sample-rows', type=int, default=0,
                    help='rows of the bounded CPU-oracle sample (default: 256; C4 128; the extra configs of the default run use 32)')
    ap.add_argument('--train-graph', action='store_true',
                    help='C5, fused trainer, one GPU: replay the step from its hipGraph (StofNetTrainer.train_step_graphed) instead of '
                         'launching kernel by kernel; measured equal (r4: 0.805 vs 0.799 ms at batch 4, 4.05 vs 4.04 ms at batch 256)')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='default run (C2, one GPU): do not measure C3 / C4 / C5 afterwards (extras.configs)')
    ap.add_argument('--dry-run', action='store_true',
                    help='rehearse the launcher and the distributed plumbing on the CPU (gloo, no kernels): tests only')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='rehearsal of the N-rank path on a one-GPU box: every rank uses cuda:0 and the collectives go over '
                         'gloo (RCCL needs one GPU per rank); the kernels and all rank-dependent code are the real ones')
    args = ap.parse_args(argv)
    if args.mode == 'train':
        args.config = 'C5'
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    return args


# ------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N  ->  N children, one per GPU.  Standard library only; the parent never
# imports torch or touches the GPU, and children are started with subprocess (never exec of a GPU process).
# ------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _tail(path, n=30):
    try:
        with open(path, 'rb') as f:
            return b''.join(f.readlines()[-n:]).decode(errors='replace')
    except OSError:
        return ''


def _stop(procs, grace=5.0):
    """End exactly the children this launcher started (by PID): SIGTERM, then SIGKILL after `grace` seconds."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        p.terminate()
    t_end = time.monotonic() + grace
    for p in live:
        try:
            p.wait(max(0.0, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def launch_children(args, argv):
    """One child per GPU; all of them are polled.  The first non-zero exit (or the watchdog) ends the siblings at once
    and the launcher returns non-zero with the failing rank's stderr tail: a dead rank cannot leave rank 0 waiting in
    RCCL init / a barrier until the process-group timeout.  Per-rank stderr goes to <logdir>/bench_rank<r>.<launcher pid>.err."""
    port = os.environ.get('MASTER_PORT') or str(free_port())
    logdir = os.environ.get('STOF_BENCH_LOGDIR') or os.path.join(ROOT, 'gpurun_out')
    os.makedirs(logdir, exist_ok=True)
    limit = float(os.environ.get('STOF_BENCH_TIMEOUT', '540'))
    procs, errs, files = [], [], []
    out0_path = os.path.join(logdir, f'bench_rank0.{os.getpid()}.out')
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        err_path = os.path.join(logdir, f'bench_rank{rank}.{os.getpid()}.err')
        ferr = open(err_path, 'wb')
        fout = open(out0_path, 'wb') if rank == 0 else subprocess.DEVNULL
        files += [ferr] + ([fout] if rank == 0 else [])
        errs.append(err_path)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=fout, stderr=ferr))
    t0 = time.monotonic()
    verdict = None
    while verdict is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            verdict = f'ranks failed (rank, exit code): {bad}'
        elif all(c == 0 for c in codes):
            verdict = ''
        elif time.monotonic() - t0 > limit:
            verdict = f'watchdog: no result after {limit:.0f} s; still running: {[r for r, c in enumerate(codes) if c is None]}'
        else:
            time.sleep(0.1)
    _stop(procs)
    for f in files:
        f.close()
    sys.stdout.write(open(out0_path).read())
    sys.stdout.flush()
    try:
        os.remove(out0_path)
    except OSError:
        pass
    if verdict:
        sys.stderr.write(f'bench.py: {verdict}\n')
        first = bad[0][0] if bad else 0
        sys.stderr.write(f'---- stderr tail of rank {first} ({errs[first]}) ----\n{_tail(errs[first])}\n')
        return 1
    for r, path in enumerate(errs):            # relay rank 0's warnings; keep the other ranks' files for inspection
        if r == 0:
            sys.stderr.write(_tail(path, 10))
    return 0


# ------------------------------------------------------------------------------------------------
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


def cpu_baseline(sd, r, L, x_np, y_gpu=None, idx_gpu=None, threshold=None, reps=3):
    """The oracle (oracle/stofnet_oracle.py, PyTorch CPU fp32; oracle/pickers_oracle.py) on a bounded sample `x_np`
    [rows,1,L]: its speed on the host cores, the batch-1 figure of BASELINE config C1 (wall and the reference's
    time.process_time() methodology, main.py:313-315), and -- as the checker -- the parity of the timed GPU outputs."""
    import numpy as np
    import torch
    from oracle import pickers_oracle as po
    from oracle import stofnet_oracle as so
    cores = host_cores()
    torch.set_num_threads(cores)
    rows = x_np.shape[0]
    with torch.no_grad():
        so.stofnet_forward(sd, x_np[:16], r)          # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            y_ref = so.stofnet_forward(sd, x_np, r)
        dt = (time.perf_counter() - t0) / reps
        b64 = None
        if rows >= 64:                                # SURVEY 8d names N = 1 and N = 64
            t0 = time.perf_counter()
            so.stofnet_forward(sd, x_np[:64], r)
            b64 = time.perf_counter() - t0
        so.stofnet_forward(sd, x_np[:1], r)
        t0, p0 = time.perf_counter(), time.process_time()
        for _ in range(5):
            so.stofnet_forward(sd, x_np[:1], r)
        w1, p1 = (time.perf_counter() - t0) / 5, (time.process_time() - p0) / 5
    out = {'value': round(rows / dt, 2), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
           'sample': f'[{rows},1,{L}] fp32, upsample_factor={r}, PyTorch-CPU oracle forward, '
                     f'{reps} reps after warm-up, torch {torch.__version__}',
           'batch1': {'waveforms_per_s': round(1.0 / w1, 2), 'wall_ms': round(w1 * 1e3, 2),
                      'process_time_ms': round(p1 * 1e3, 2),
                      'note': 'C1 shape [1,1,L]; process_time = CPU time over all threads, the reference\'s main.py:313-315 method'}}
    if b64 is not None:
        out['batch64'] = {'waveforms_per_s': round(64.0 / b64, 2), 'wall_ms': round(b64 * 1e3, 1)}
    if y_gpu is not None:
        y_ref = y_ref.numpy()
        prow = min(rows, 64 if threshold else rows)
        ref = po.mask2coords(y_ref[:prow], 20, threshold, 1)                     # [prow, K] (zero padded) in output samples
        got = np.zeros_like(ref) if ref.ndim == 2 else ref
        if ref.ndim == 2:
            k = min(ref.shape[1], idx_gpu.shape[1])
            g = idx_gpu[:prow, :k].cpu().numpy().astype(np.float32)
            got[:, :k] = g
        out['parity_on_sample'] = {
            'picker': 'arg-max' if not threshold else f'threshold {threshold}', 'rows': int(prow),
            'onset_index_mae': float(np.abs(got - ref).mean()) if ref.ndim == 2 else 0.0,
            'onset_index_mismatches': int((got != ref).sum()) if ref.ndim == 2 else 0,
            'max_rel_err_maps': float(np.abs(y_gpu[:rows].cpu().numpy() - y_ref).max() / np.abs(y_ref).max())}
    return out


class Dist:
    """torch.distributed plumbing shared by every config: barrier, MAX of the timed interval."""

    def __init__(self, args, backend):
        import torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch with '
                             f'`python bench.py --gpus N` or torch.distributed.run --nproc-per-node N')
        self.dist = None
        self.rehearsal = bool(getattr(args, 'rehearse_on_one_gpu', False)) and backend == 'nccl'
        if self.rehearsal:
            backend, self.local_rank = 'gloo-gpu', 0
        self.backend = backend
        self.dev = torch.device('cpu') if backend == 'gloo' else torch.device('cuda', self.local_rank)
        fail_rank = os.environ.get('STOF_TEST_FAIL_RANK')          # tests/test_bench_launcher.py: one rank dies before init
        if fail_rank is not None and int(fail_rank) == self.rank:
            sys.stderr.write(f'bench.py: rank {self.rank} exits 3 on purpose (STOF_TEST_FAIL_RANK)\n')
            sys.stderr.flush()
            os._exit(3)
        if os.environ.get('STOF_TEST_HANG_RANK') is not None and int(os.environ['STOF_TEST_HANG_RANK']) == self.rank:
            time.sleep(3600)                                       # tests: the launcher's watchdog must end this
        if backend == 'nccl' and torch.cuda.device_count() < max(self.world, self.local_rank + 1):
            raise SystemExit(f'bench.py: rank {self.rank} sees {torch.cuda.device_count()} GPU(s) but WORLD_SIZE={self.world} '
                             f'(one process per GPU): refusing to start')
        # STOF_FORCE_PROCESS_GROUP=1: create the process group even for one rank, so a one-GPU box runs the RCCL path
        # (communicator set-up, int32 all_gather, MAX / SUM all-reduces) that an 8-GPU node would (tests/test_rccl_one_rank.py)
        if self.world > 1 or os.environ.get('STOF_FORCE_PROCESS_GROUP') == '1':
            import datetime
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            tmo = datetime.timedelta(seconds=float(os.environ.get('STOF_DIST_TIMEOUT', '180')))
            if backend == 'nccl':
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.dev, timeout=tmo)
            else:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world, timeout=tmo)
            self.dist = dist
        if backend != 'gloo':
            torch.cuda.set_device(self.dev)

    def census(self):
        """Proof in the record that `world` distinct devices took part: ranks_seen = SUM all-reduce of ones, and the
        gathered identity (uuid / PCI bus id / name) of every rank's device.  Rehearsal runs say so in the line."""
        import torch
        ident = {'rank': self.rank, 'device': str(self.dev)}
        if self.dev.type == 'cuda':
            pr = torch.cuda.get_device_properties(self.dev)
            ident['name'] = pr.name
            for key in ('uuid', 'pci_bus_id', 'pci_device_id', 'pci_domain_id'):
                if hasattr(pr, key):
                    ident[key] = str(getattr(pr, key))
        seen, idents = 1, [ident]
        if self.dist is not None:
            one = torch.ones(1, dtype=torch.int32, device=self.dev)
            self.dist.all_reduce(one, op=self.dist.ReduceOp.SUM)
            seen = int(one.item())
            idents = [None] * self.world
            self.dist.all_gather_object(idents, ident)
        keys = {(i.get('uuid'), i.get('pci_domain_id'), i.get('pci_bus_id'), i.get('pci_device_id'))
                if (i.get('uuid') or i.get('pci_bus_id')) else i['device'] for i in idents}
        physical = 1 if (self.rehearsal or self.dev.type != 'cuda') else len(keys)
        out = {'ranks_seen': seen, 'physical_gpus': physical if self.dev.type == 'cuda' else 0,
               'collective_backend': {'nccl': 'nccl (RCCL)', 'gloo-gpu': 'gloo (one-GPU rehearsal)', 'gloo': 'gloo (CPU dry run)'}[self.backend],
               'devices': idents}
        if self.rehearsal:
            out['rehearsal_on_one_gpu'] = True
        if self.dev.type == 'cuda' and not self.rehearsal and physical != self.world:
            out['warning'] = f'{self.world} ranks but {physical} distinct device identities'      # reported, not fatal
        return out

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync(self):
        import torch
        if self.dev.type == 'cuda':
            torch.cuda.synchronize()

    def timed(self, fn, steps):
        """EXACTLY `steps` calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks."""
        import torch
        self.sync()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for s in range(steps):
            fn(s)
        self.sync()
        self.barrier()
        self.sync()
        dt = time.perf_counter() - t0
        if self.dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def base_line(args, d, metric, value, dt, dtype, workload, config_extra):
    return {'metric': metric, 'value': round(value, 1), 'unit': 'waveforms/s', 'n_gpus': d.world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': dict({'workload': workload}, **config_extra)}


def dry_run(args):
    """Launcher / process-group rehearsal on the CPU (gloo): the same barrier + max-over-ranks timing and the same
    ragged index gather as the real configs, with a no-op step.  Used by tests/test_bench_launcher.py."""
    import torch
    from stofnet_amd.sharding import gather_onsets
    d = Dist(args, 'gloo')
    rows = args.rows or 8
    dt = d.timed(lambda s: time.sleep(0.001), args.steps)
    counts = torch.full((rows,), 1 + d.rank, dtype=torch.int32)
    idx = torch.arange(rows * (1 + d.rank), dtype=torch.int32).reshape(rows, 1 + d.rank) + 1000 * d.rank
    gather_ms = None
    if d.dist is not None:
        t1 = time.perf_counter()
        counts_all, idx_all = gather_onsets(counts, idx)
        gather_ms = (time.perf_counter() - t1) * 1e3
        assert counts_all.shape[0] == rows * d.world and idx_all.shape[1] == d.world
    census = d.census()
    if d.rank == 0:
        out = base_line(args, d, 'dry run (no kernels)', d.world * rows * args.steps / dt, dt, 'none',
                        'launcher rehearsal on CPU/gloo', {'rows_per_gpu': rows})
        out['dry_run'] = True
        out['ranks'] = census
        out['extras'] = {'index_gather_ms': None if gather_ms is None else round(gather_ms, 3)}
        print(json.dumps(out), flush=True)
    d.finish()


def train_bench(args):
    """BASELINE.json configs[4]: one training step = forward (activations kept) + Gaussian-mask loss + backward +
    mean all-reduce of the single 2.58 MB gradient bucket (RCCL, N > 1) + AdamW, batch per GPU fixed."""
    import numpy as np
    import torch
    from stofnet_amd import StofNet, synth
    from stofnet_amd.training import StofNetTrainer
    d = Dist(args, 'nccl')
    R, L = 10, L_CHIRP
    sd = synth.synth_state_dict(R, seed=3008)
    model = StofNet(upsample_factor=R)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(d.dev)
    tr = StofNetTrainer(model, precision=args.train_precision) if args.trainer == 'fused' else None
    nb = args.rows or args.train_batch
    x = torch.from_numpy(synth.synth_echo(nb, L, seed=3008 + d.rank)).to(d.dev)
    rng = np.random.default_rng(d.rank)
    gt = torch.from_numpy(np.sort(rng.integers(1, L * R, size=(nb, 1, 2)), -1)).to(d.dev)
    last = {}

    # --train-graph: one hipGraph per step shape (forward + loss + backward + range guard; AdamW behind it): the same kernels
    graphed = args.trainer == 'fused' and d.world == 1 and args.train_graph

    def step(_s):
        last['loss'], _ = (tr.train_step_graphed if graphed else tr.train_step)(x, gt)

    if args.trainer == 'autograd':
        import torch.nn.functional as F
        from stofnet_amd.mask2samples import coords2mask
        from stofnet_amd.training import allreduce_max_, allreduce_mean_, gaussian_kernel
        model.train_precision = args.train_precision
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-8)
        gauss = torch.tensor(gaussian_kernel(7, 1.0), dtype=torch.float32, device=d.dev).unsqueeze(0).unsqueeze(0)
        mse, l1 = torch.nn.MSELoss(reduction='mean'), torch.nn.L1Loss(reduction='mean')

        def step(_s):      # noqa: F811  -- the reference's lines, main.py:221-248
            pred = model(x)
            blur = F.conv1d(coords2mask(gt.clone(), pred), gauss, padding=3)
            blur = blur / allreduce_max_(blur.max().reshape(1)) * 20
            loss = mse(pred.squeeze(1), blur.squeeze(1).float()) + l1(pred.squeeze(1), torch.zeros_like(pred.squeeze(1))) * 1e-2
            opt.zero_grad()
            loss.backward()
            if d.world > 1:
                for prm in model.parameters():
                    allreduce_mean_(prm.grad)
            opt.step()
            last['loss'] = loss.detach()

    for s in range(args.warmup):
        step(s)
    dt = d.timed(step, args.steps)
    census = d.census()
    d.finish()                                      # all collectives done: rank 0's CPU work below cannot time the group out
    if d.rank == 0:
        flops = 3.0 * total_flops(nb, L, R)        # forward + data-gradient + weight-gradient
        achieved = flops * args.steps / dt / 1e12
        peak = PEAK_TFLOPS[args.train_precision]
        out = base_line(args, d, 'RF waveforms/sec StofNet training step (fwd+bwd+AdamW) rf_scale=10',
                        d.world * nb * args.steps / dt, dt, DTYPE_TEXT[args.train_precision],
                        f'C5 training step [{nb},1,{L}] -> [{nb},1,{L * R}] per GPU, Gaussian-mask loss, AdamW, upsample_factor={R}',
                        {'rows_per_gpu': nb, 'L': L, 'upsample_factor': R, 'precision': args.train_precision,
                         'trainer': args.trainer, 'hip_graph': bool(graphed),
                         'parallelism': f'ddp{d.world}: one flat 2.58 MB gradient all-reduce per step'})
        # `achieved` counts the ALGORITHMIC flops of the step (what the reference's dense autograd does).  Since r3 the two backward
        # passes of the SemiGlobalBlock's contract convolution (22.6 % of that count) work on the max-pool's sparse gradient and are
        # not multiplied out; `executed` leaves them out, i.e. it is the matrix-pipe work actually done per second.
        eng = tr if tr is not None else model._engine(d.dev, args.train_precision)
        sparse_sgb = bool(getattr(eng, 'sgb_sparse_taken', False))      # what the engine's last backward really ran
        executed = (flops - (2.0 * 2.0 * nb * L * 163840 if sparse_sgb else 0.0)) * args.steps / dt / 1e12
        # `achieved` / `frac` = the matrix-pipe work actually EXECUTED per second; the algorithmic count (what the reference's dense
        # autograd multiplies out, 22.6 % more when the sparse SemiGlobalBlock backward is taken) is given beside it
        out['roofline'] = {'bound': 'mfma', 'kernel': f'whole step (sweeps + conv_wgrad / conv_cl16 kernels, {args.train_precision} MFMA)',
                           'achieved': round(executed, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                           'frac': round(executed / peak, 4), 'traffic': None,
                           'algorithmic': round(achieved, 2), 'algorithmic_frac': round(achieved / peak, 4),
                           'sparse_sgb_backward_taken': sparse_sgb}
        out['final_loss'] = float(last['loss'])
        out['ranks'] = census
        if not args.no_cpu_baseline and d.world == 1:       # rank 0 at N = 1 only (the process group is already closed)
            # the oracle's training step (torch autograd on the host cores), bounded sample
            from oracle import train_oracle
            cores = host_cores()
            torch.set_num_threads(cores)
            rows = min(args.cpu_sample_rows or 32, nb)
            xs, gts = x[:rows].cpu().numpy(), gt[:rows].cpu().numpy()
            train_oracle.loss_and_grads(sd, xs[:4], gts[:4], R, 80, dtype=torch.float32)
            t1 = time.perf_counter()
            train_oracle.loss_and_grads(sd, xs, gts, R, 80, dtype=torch.float32)
            cdt = time.perf_counter() - t1
            out['cpu_baseline'] = {'value': round(rows / cdt, 2), 'unit': 'waveforms/s', 'cores': cores, 'kind': 'port',
                                   'sample': f'1 fwd+bwd of {rows} waveforms (oracle, torch autograd fp32), optimizer step excluded'}
            if tr is not None:
                # the checker: loss and every parameter gradient of the engine, at its CURRENT weights, against the oracle's
                # autograd (float64) on the first rows of the batch
                prow = min(8, nb)
                cur = {k: v.detach().cpu().numpy().copy() for k, v in tr.p.items()}
                gl, _ = tr.forward_backward(x[:prow], gt[:prow])
                g_gpu = {k: v.detach().cpu().numpy().copy() for k, v in tr.g.items()}
                ol, og, _ = train_oracle.loss_and_grads(cur, x[:prow].cpu().numpy(), gt[:prow].cpu().numpy(), R, 80, dtype=torch.float64)
                worst = max(float(np.abs(g_gpu[k] - og[k]).max() / max(np.abs(og[k]).max(), 1e-30)) for k in og)
                out['parity_on_sample'] = {'rows': prow, 'loss_rel_err': abs(float(gl) - ol) / abs(ol),
                                           'max_grad_err_rel_to_max_abs_grad': worst,
                                           'note': 'engine (weights after the timed steps) vs oracle autograd in float64'}
        print(json.dumps(out), flush=True)


def run_extra_configs(args):
    """The other BASELINE configs under the same clock as the default (C2) line: each is this script again as a CHILD process
    (started with subprocess after the C2 timed region -- never an exec of this GPU process), a short run with a small CPU
    sample; its line is distilled to {value, ms_per_step, roofline, parity_on_sample, ...}."""
    runs = [('C3', ['--config', 'C3', '--steps', '5', '--warmup', '1']),
            ('C4', ['--config', 'C4', '--steps', '2', '--warmup', '1']),
            ('C5', ['--config', 'C5', '--steps', '50', '--warmup', '5']),
            ('C5_batch4', ['--config', 'C5', '--steps', '50', '--warmup', '5', '--rows', '4'])]
    out = {}
    for name, extra in runs:
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--no-extra-configs', '--no-fp32-extra',
               '--cpu-sample-rows', '32', '--precision', args.precision, '--train-precision', args.train_precision] + extra
        t0 = time.perf_counter()
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=float(os.environ.get('STOF_EXTRA_TIMEOUT', '150')))
            line = [ln for ln in pr.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
            if pr.returncode != 0 or not line:
                out[name] = {'error': f'exit code {pr.returncode}', 'stderr_tail': pr.stderr.decode(errors='replace')[-400:]}
                continue
            rec = json.loads(line[-1])
        except subprocess.TimeoutExpired:
            out[name] = {'error': 'timed out'}
            continue
        keep = {k: rec[k] for k in ('metric', 'value', 'unit', 'steps', 'ms_per_step', 'dtype', 'roofline', 'kernels_ms', 'final_loss',
                                    'parity_on_sample') if k in rec}
        keep['workload'] = rec['config']['workload']
        if 'cpu_baseline' in rec:
            keep['cpu_baseline'] = {k: rec['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'sample') if k in rec['cpu_baseline']}
            if 'parity_on_sample' in rec['cpu_baseline']:
                keep['parity_on_sample'] = rec['cpu_baseline']['parity_on_sample']
        ex = rec.get('extras', {})
        for k in ('forward_only_ms_per_step', 'picker_threshold_ms_per_step', 'detections_per_row_mean', 'kmax', 'fused_argmax_onsets'):
            if k in ex:
                keep[k] = ex[k]
        keep['child_wall_s'] = round(time.perf_counter() - t0, 1)
        out[name] = keep
    return out


def load_fixture_weights(key):
    """Checkpoint tensors committed as data under tests/golden (the reference's .pth cannot travel)."""
    import numpy as np
    d = np.load(os.path.join(ROOT, 'tests', 'golden', f'weights_{key}.npz'))
    return {k: d[k] for k in d.files}


def infer_bench(args):
    import ctypes
    import numpy as np
    import torch
    from stofnet_amd import StofNet, _lib, synth
    from stofnet_amd.mask2samples import onset_indices, pick_async
    from stofnet_amd.sharding import gather_onsets
    d = Dist(args, 'nccl')
    dev = d.dev
    cfg = args.config
    if cfg == 'C4':
        R, L, chunk = C4_R, C4_L, C4_CHUNK
        rows = args.rows or C4_ROWS
        chunk = min(chunk, rows)
        rows = rows // chunk * chunk
        sd = load_fixture_weights('graceful-snow')
        th = C4_TH
        # 8192 host-generated echoes (seeded, NormalizeVol), replicated on the device with fresh seeded noise per chunk
        base = torch.from_numpy(synth.synth_echo(chunk, L, seed=3008 + d.rank)).to(dev)
        g = torch.Generator(device=dev)
        g.manual_seed(3008 + d.rank)
        x = torch.empty((rows, 1, L), dtype=torch.float32, device=dev)
        for c in range(rows // chunk):
            v = base if c == 0 else base + 0.01 * torch.randn(base.shape, generator=g, device=dev)
            x[c * chunk:(c + 1) * chunk] = v / v.abs().amax(dim=-1, keepdim=True)
        workload = (f'C4 PALA shape: {rows} rows x {L} samples per GPU in {chunk}-row chunks, StofNet.forward -> '
                    f'[{chunk},1,{L * R}] + mask2coords picker (threshold {th}, window 20), upsample_factor={R}, '
                    f'checkpoint graceful-snow')
    else:
        R = 20 if cfg == 'C3' else 10
        L, rows = L_CHIRP, args.rows or 4096
        chunk, th = rows, None
        sd = synth.synth_state_dict(R, seed=3008)
        x = torch.from_numpy(synth.synth_randn(rows, L, seed=3008 + d.rank)).to(dev)   # resident before timing
        workload = (f'{cfg} StofNet.forward [{rows},1,{L}] -> [{rows},1,{L * R}] per GPU, upsample_factor={R}, '
                    f'SemiGlobalBlock on, seeded-random weights (seed 3008)')
    nchunk = rows // chunk
    model = StofNet(upsample_factor=R, precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()

    lib = _lib.lib()
    nev = 4
    ev_sets = []
    for _ in range(args.steps):
        arr = (ctypes.c_void_p * nev)()
        _lib.check(lib.stof_events_create(nev, arr))
        ev_sets.append(arr)

    PICK_CAP = 256          # detections kept per row (threshold mode on noisy rows finds tens); checked after the timed region
    res = {}
    if cfg == 'C4':
        counts_all = torch.zeros((rows,), dtype=torch.int32, device=dev)
        idx_all = torch.zeros((rows, PICK_CAP), dtype=torch.int32, device=dev)

    def step(s, events=True):
        for c in range(nchunk):
            xs = x if nchunk == 1 else x[c * chunk:(c + 1) * chunk]
            y = model(xs, _events=ev_sets[s] if (events and c == 0) else None)
            if cfg == 'C4':                   # the C4 hot path includes the picker; no host sync inside the step
                pick_async(y, 20, th, counts=counts_all[c * chunk:(c + 1) * chunk], idx=idx_all[c * chunk:(c + 1) * chunk])
        res['y'] = y

    step(0, events=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(0, events=False)
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-4)
    preheat = max(1, int(PREHEAT_SECONDS / one))
    for _ in range(preheat):                        # untimed: lets the clocks/power state settle whatever W is
        step(0, events=False)
    for _ in range(args.warmup):
        step(0, events=False)
    dt = d.timed(step, args.steps)
    y = res['y']
    fell_back = model.fell_back_to_fp32() if args.precision == 'auto' else False
    if args.precision == 'f16x3':
        model.raise_if_overflow()

    # per-kernel durations from the HIP events recorded inside the timed region (first chunk of every step)
    kern = np.zeros((args.steps, 3))
    ms = ctypes.c_float()
    for s, arr in enumerate(ev_sets):
        for k in range(3):
            _lib.check(lib.stof_event_elapsed_ms(arr[k], arr[k + 1], ctypes.byref(ms)))
            kern[s, k] = ms.value
        lib.stof_events_destroy(nev, arr)
    k_ms = kern.mean(0)

    # extras outside the timed region: picker and optional RCCL gather of the onset indices
    extras = {}
    if cfg == 'C4':
        kmax = int(counts_all.max())
        if kmax > PICK_CAP:
            raise SystemExit(f'bench.py: a row produced {kmax} detections > cap {PICK_CAP}')
        counts = counts_all
        ar = torch.arange(max(kmax, 1), device=dev)[None, :]
        idx = torch.where(ar < counts[:, None], idx_all[:, :max(kmax, 1)], torch.zeros((), dtype=torch.int32, device=dev)).contiguous()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for c in range(nchunk):
            model(x[c * chunk:(c + 1) * chunk])
        torch.cuda.synchronize()
        fwd_only = time.perf_counter() - t1
        extras['forward_only_ms_per_step'] = round(fwd_only * 1e3, 3)
        extras['picker_threshold_ms_per_step'] = round(dt / args.steps * 1e3 - fwd_only * 1e3, 3)
        extras['detections_per_row_mean'] = round(float(counts.float().mean()), 3)
        extras['kmax'] = kmax
    else:
        counts, idx = onset_indices(y, 20, None)          # warm-up: first use loads the picker's code object
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            counts, idx = onset_indices(y, 20, None)      # includes the Kmax host sync the reference has too
        torch.cuda.synchronize()
        pick_ms = (time.perf_counter() - t1) * 1e3 / 5
        ar = torch.arange(idx.shape[1], device=dev)[None, :]
        idx = torch.where(ar < counts[:, None], idx, torch.zeros((), dtype=torch.int32, device=dev)).contiguous()
        extras['picker_argmax_ms'] = round(pick_ms, 3)
        extras['waveforms_per_s_forward_plus_picker'] = round(d.world * rows / (dt / args.steps + pick_ms * 1e-3), 1)
    # the arg-max picker fused into the sweep (stof_forward_onsets): picker-only output, the map never reaches HBM
    if args.precision != 'fp32' and R <= 16:
        def fused_pass():
            for c in range(nchunk):
                model.forward_onsets(x if nchunk == 1 else x[c * chunk:(c + 1) * chunk], 20)
        def fused_stream():      # serving form: no host read per call; counts / guard word are read after the loop
            for c in range(nchunk):
                res['fused'] = model.forward_onsets(x if nchunk == 1 else x[c * chunk:(c + 1) * chunk], 20, sync=False)
        fused_pass()
        fused_stream()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            fused_pass()
        torch.cuda.synchronize()
        fdt = (time.perf_counter() - t1) / 3
        t1 = time.perf_counter()
        for _ in range(5):
            fused_stream()
        torch.cuda.synchronize()
        fdt_ns = (time.perf_counter() - t1) / 5
        kmax_ns = int(res['fused'][0].max())
        stream_overflow = bool(model.onsets_overflowed())            # sticky range-guard word of ALL sync=False calls above
        stream_truncated = bool((res['fused'][0] > res['fused'][1].shape[1]).any())
        if stream_overflow or stream_truncated:
            raise SystemExit(f'bench.py: fused onsets stream invalid (fp16 range overflow: {stream_overflow}, '
                             f'row list truncated: {stream_truncated})')
        fc, fi = model.forward_onsets(x[:chunk], 20)
        rc, ri = onset_indices(model(x[:chunk]), 20, None)
        extras['fused_argmax_onsets'] = {'ms_per_step': round(fdt_ns * 1e3, 3), 'waveforms_per_s': round(d.world * rows / fdt_ns, 1),
                                         'ms_per_step_with_host_sync': round(fdt * 1e3, 3),
                                         'waveforms_per_s_with_host_sync': round(d.world * rows / fdt, 1),
                                         'kmax_read_after_loop': kmax_ns,
                                         'range_guard_sticky_word_after_loop': int(stream_overflow),
                                         'rows_truncated_after_loop': int(stream_truncated),
                                         'identical_to_map_plus_picker': bool(torch.equal(fc, rc) and torch.equal(fi, ri)),
                                         'note': 'StofNet.forward_onsets(sync=False): output = onset indices only, counts and the '
                                                 'range-guard word read once after the timed loop; the *_with_host_sync figures '
                                                 'include the per-call Kmax read the reference has (utils/mask2samples.py:93)'}
    extras['index_gather_ms'] = None
    if d.dist is not None:
        gather_onsets(counts, idx)                       # warm-up (communicator set-up)
        torch.cuda.synchronize()
        d.barrier()
        t1 = time.perf_counter()
        c_all, i_all = gather_onsets(counts, idx)
        torch.cuda.synchronize()
        extras['index_gather_ms'] = round((time.perf_counter() - t1) * 1e3, 3)
        extras['gathered_rows'] = int(c_all.shape[0])
        extras['gathered_kmax'] = int(i_all.shape[1])

    # secondary measurement: the exact-fp32 parity-baseline mode, same workload (outside the timed region)
    if args.precision != 'fp32' and not args.no_fp32_extra and cfg != 'C4':
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
        extras['fp32_exact_mode'] = {'waveforms_per_s_per_gpu': round(rows / dt32, 1), 'ms_per_step': round(dt32 * 1e3, 3),
                                     'max_rel_diff_vs_timed_mode': float((y32 - y).abs().max() / y32.abs().max()),
                                     'onset_index_mismatches_vs_timed_mode': int((i32[:, :1] != idx[:, :1]).sum())}
        del m32, y32
    extras['auto_mode_fp32_rerun_taken'] = bool(fell_back)
    census = d.census()
    # every collective is over: the ranks part here, so that rank 0's CPU baseline (tens of seconds on the host cores) cannot
    # run into the process group's timeout while the others wait in a barrier
    d.finish()

    if d.rank == 0:
        launch_rows = min(chunk, 4096)              # stof_forward sweeps sub-batches of <= 4096 rows per launch
        body_s = k_ms[2] * 1e-3
        achieved = body_flops(launch_rows, L, R) / body_s / 1e12
        peak = PEAK_TFLOPS[args.precision]
        # HBM bytes per launch cannot be read inside this process (rocprofv3 PMC passes run the whole command under the
        # profiler, tools/profile_bench.sh): the figure is the committed record of such a pass and is labelled as one
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath) and cfg == 'C2' and rows == 4096 and args.precision != 'fp32':
            try:
                rec = json.load(open(tpath))
                traffic = rec.get('body_sweep_hbm_bytes_per_launch')
                traffic_source = (f"NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, "
                                  f"profiles/traffic_latest.json <- {rec.get('source')}; recorded {rec.get('date', 'round 2')}")
            except Exception:  # noqa: BLE001
                traffic, traffic_source = None, None
        metric = 'RF waveforms/sec StofNet inference rf_scale=10'
        if cfg == 'C3':
            metric = 'RF waveforms/sec StofNet inference rf_scale=20'
        if cfg == 'C4':
            metric = 'RF waveforms/sec StofNet inference + onset picker, PALA shape'
        out = base_line(args, d, metric, d.world * rows * args.steps / dt, dt, DTYPE_TEXT[args.precision], workload,
                        {'rows_per_gpu': rows, 'L': L, 'upsample_factor': R, 'sharding': f'batch x{d.world}, no collective',
                         'precision': args.precision})
        out['preheat_steps'] = preheat
        out['roofline'] = {'bound': 'mfma', 'kernel': 'body_sweep_kernel', 'achieved': round(achieved, 2),
                           'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                           'traffic_source': traffic_source, 'flops_per_launch': body_flops(launch_rows, L, R), 'avg_launch_ms': round(float(k_ms[2]), 4),
                           'rows_per_launch': launch_rows}
        out['kernels_ms'] = {'sgb_contract_pool': round(float(k_ms[0]), 4), 'sgb_expand': round(float(k_ms[1]), 4),
                             'body_sweep': round(float(k_ms[2]), 4)}
        out['whole_forward_tflops'] = round(total_flops(rows, L, R) * d.world * args.steps / dt / 1e12, 2)
        out['extras'] = extras
        out['ranks'] = census
        if not args.no_cpu_baseline and d.world == 1:       # rank 0 at N = 1 only (the process group is already closed)
            srows = args.cpu_sample_rows or (256 if cfg != 'C4' else 128)
            srows = min(srows, rows)
            xs = x[:srows].cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(sd, R, L, xs, y[:srows] if cfg != 'C4' else model(x[:srows]),
                                               idx[:srows], threshold=th)
        if cfg == 'C2' and d.world == 1 and not args.no_extra_configs and not args.rows:
            # BASELINE.json configs[2..4] under the same clock, after the headline's timed region (headline keys unchanged)
            del y
            torch.cuda.empty_cache()
            extras['configs'] = run_extra_configs(args)
        print(json.dumps(out), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        return launch_children(args, argv)
    if args.dry_run:
        return dry_run(args)
    if args.config == 'C5':
        return train_bench(args)
    return infer_bench(args)


if __name__ == '__main__':
    sys.exit(main() or 0)
