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
