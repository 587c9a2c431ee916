#!/bin/bash
# last checks of the round: the whole GPU suite, then the 2-rank path rehearsed on one GPU (C2 and C5)
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_tests.log 2>&1 || { tail -40 gpurun_out/r04_final_tests.log; exit 1; }
tail -2 gpurun_out/r04_final_tests.log
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-extra > gpurun_out/r04_rehearse2_c2.json 2> gpurun_out/r04_rehearse2_c2.err || { tail -20 gpurun_out/r04_rehearse2_c2.err; exit 1; }
cut -c1-300 gpurun_out/r04_rehearse2_c2.json
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --config C5 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r04_rehearse2_c5.json 2> gpurun_out/r04_rehearse2_c5.err || { tail -20 gpurun_out/r04_rehearse2_c5.err; exit 1; }
cut -c1-300 gpurun_out/r04_rehearse2_c5.json
