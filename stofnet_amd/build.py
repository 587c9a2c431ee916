"""Build libstofnet_amd.so (gfx950 code objects + C ABI) in-tree with hipcc.

    python -m stofnet_amd.build          # or __graft_entry__.build()

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels
to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libstofnet_amd.so')
SOURCES = ['pack_weights.cpp', 'convstack.hip', 'shuffle_picker.hip', 'hilbert.hip', 'gradpeak.hip', 'neighbors.hip', 'train.hip']
ARCH = 'gfx950'
# convstack.hip: the SLP vectoriser pairs scalar fp32 FMAs into v_pk_fma_f32 (slow beside MFMAs) and thereby defeats the
# v_fma_mix_f32 selection of the sweep's epilogue (see mix_add in convstack.hip)
EXTRA_FLAGS = {'convstack.hip': ['-fno-slp-vectorize']}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP extension cannot be built')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h'))] + [os.path.join(HERE, '..', 'include', 'stofnet_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            raise RuntimeError(f'listed source {path} is missing: refusing to link a library without its symbols')
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        cmd = [hipcc, '-O3', '-std=c++17', '-fPIC', '-fconstexpr-steps=100000000', f'--offload-arch={ARCH}', '-x', 'hip'] + \
              EXTRA_FLAGS.get(src, []) + ['-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
