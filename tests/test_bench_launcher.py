"""CPU rehearsal of bench.py's multi-rank launch (`python bench.py --gpus N`): the parent spawns N children, rank 0
prints ONE JSON line with n_gpus = N, the barrier + max-over-ranks timing and the ragged index gather run over gloo.
No kernels run here (--dry-run); the GPU box runs the same plumbing over RCCL."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, 'bench.py')


def run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith('{')]


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = run(['--gpus', '2', '--steps', '4', '--warmup', '1', '--dry-run', '--rows', '5'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['extras']['index_gather_ms'] is not None          # the ragged gather ran across both ranks
    assert out['config']['rows_per_gpu'] == 5
    assert abs(out['value'] - 2 * 5 * 4 / (out['ms_per_step'] * 4e-3)) / out['value'] < 1e-3   # whole-job aggregate


def test_single_rank_needs_no_process_group():
    p = run(['--steps', '2', '--warmup', '0', '--dry-run'])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json_lines(p.stdout)[0]
    assert out['n_gpus'] == 1 and out['extras']['index_gather_ms'] is None


def test_gpus_flag_must_agree_with_world_size():
    """Launched by torch.distributed.run the ranks come from the environment; a mismatch fails loudly instead of
    reporting a wrong n_gpus."""
    p = run(['--gpus', '2', '--steps', '1', '--dry-run'], env={'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1'})
    assert p.returncode != 0
    assert 'WORLD_SIZE=1' in (p.stderr + p.stdout)


def test_failed_rank_gives_nonzero_exit():
    p = run(['--gpus', '2', '--steps', '1', '--dry-run', '--config', 'bogus'])
    assert p.returncode != 0


def test_one_dead_rank_ends_the_run_quickly(tmp_path):
    """Rank 1 dies before the rendezvous (STOF_TEST_FAIL_RANK): rank 0 would wait in init_process_group until the
    process-group timeout; the launcher polls every child, ends the survivors and reports the dead rank."""
    import time
    t0 = time.monotonic()
    p = run(['--gpus', '2', '--steps', '1', '--dry-run'], env={'STOF_TEST_FAIL_RANK': '1', 'STOF_BENCH_LOGDIR': str(tmp_path)},
            timeout=120)
    took = time.monotonic() - t0
    assert p.returncode != 0
    assert took < 30, f'launcher took {took:.1f} s to notice a dead rank'
    assert '(1, 3)' in p.stderr and 'STOF_TEST_FAIL_RANK' in p.stderr         # which rank, which code, its stderr tail
    assert json_lines(p.stdout) == []


def test_watchdog_ends_a_hung_run(tmp_path):
    p = run(['--gpus', '2', '--steps', '1', '--dry-run'], env={'STOF_TEST_HANG_RANK': '1', 'STOF_BENCH_TIMEOUT': '8',
                                                                'STOF_BENCH_LOGDIR': str(tmp_path)}, timeout=120)
    assert p.returncode != 0 and 'watchdog' in p.stderr


def test_line_proves_the_ranks_that_took_part(tmp_path):
    p = run(['--gpus', '2', '--steps', '2', '--warmup', '0', '--dry-run'], env={'STOF_BENCH_LOGDIR': str(tmp_path)})
    assert p.returncode == 0, p.stderr[-2000:]
    ranks = json_lines(p.stdout)[0]['ranks']
    assert ranks['ranks_seen'] == 2 and [r['rank'] for r in ranks['devices']] == [0, 1]
    assert ranks['physical_gpus'] == 0 and 'gloo' in ranks['collective_backend']      # a CPU dry run says so in the record


def test_under_torch_distributed_run_like_the_driver(tmp_path):
    """The driver's N > 1 launch line: torch.distributed.run provides RANK / WORLD_SIZE / MASTER_*."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = dict(os.environ, STOF_BENCH_LOGDIR=str(tmp_path))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--dry-run'], capture_output=True, text=True, env=e, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['ranks']['ranks_seen'] == 2
