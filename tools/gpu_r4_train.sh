#!/bin/bash
# r4: batched 64->64 weight gradients, deterministic conv1 weight gradient, deferred range guard -- tests + C5 A/B
set -o pipefail
mkdir -p gpurun_out/r4t
O=gpurun_out/r4t
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_rccl_one_rank.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
B="python bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 50 --warmup 5"
timeout -k 10 200 $B > $O/c5_batch.json 2>$O/c5_batch.err && tail -1 $O/c5_batch.json | cut -c1-400 &&
STOF_TRAIN_WGRAD_BATCH=0 timeout -k 10 200 $B > $O/c5_nobatch.json 2>$O/c5_nobatch.err && tail -1 $O/c5_nobatch.json | cut -c1-400 &&
timeout -k 10 200 $B --trainer autograd > $O/c5_autograd.json 2>$O/c5_autograd.err && tail -1 $O/c5_autograd.json | cut -c1-400 &&
timeout -k 10 200 $B --rows 4 > $O/c5_b4.json 2>$O/c5_b4.err && tail -1 $O/c5_b4.json | cut -c1-300 &&
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && ls $O/prof | head
