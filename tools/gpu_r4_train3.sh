#!/bin/bash
set -o pipefail
O=gpurun_out/r4t3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
B="python bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 50 --warmup 5"
timeout -k 10 200 $B > $O/c5.json 2>$O/c5.err && tail -1 $O/c5.json | cut -c1-200 &&
STOF_LIB_PATH=stofnet_amd/libstof_stamps.so timeout -k 10 300 python tools/read_train_stamps.py > $O/stamps.txt 2>&1 && cat $O/stamps.txt
