#!/bin/bash
# Rehearsal of the N-rank bench path on a ONE-GPU box: both ranks run the real kernels on cuda:0, the collectives
# (barrier, MAX of the step time, index gather, gradient all-reduce) go over gloo because RCCL wants one GPU per rank.
# The figures are NOT throughput claims (two ranks share one GPU); the run shows that every rank-dependent code path
# of bench.py executes on the hardware.  Output: gpurun_out/r03_rehearse_ranks_one_gpu.jsonl
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
out=gpurun_out/r03_rehearse_ranks_one_gpu.jsonl
: > $out
python bench.py --gpus 2 --steps 5 --warmup 1 --rows 1024 --rehearse-on-one-gpu --no-fp32-extra 2>> gpurun_out/rehearse.err >> $out
python bench.py --gpus 2 --config C3 --steps 3 --warmup 1 --rows 512 --rehearse-on-one-gpu --no-fp32-extra 2>> gpurun_out/rehearse.err >> $out
python bench.py --gpus 2 --config C4 --rows 16384 --steps 2 --warmup 1 --rehearse-on-one-gpu 2>> gpurun_out/rehearse.err >> $out
python bench.py --gpus 2 --config C5 --rows 64 --steps 3 --warmup 1 --rehearse-on-one-gpu 2>> gpurun_out/rehearse.err >> $out
python bench.py --gpus 4 --steps 3 --warmup 1 --rows 512 --rehearse-on-one-gpu --no-fp32-extra 2>> gpurun_out/rehearse.err >> $out
wc -l $out
