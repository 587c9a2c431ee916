#!/bin/bash
# r4: weight-gradient kernel variants -- training tests + C5 line + kernel stats
set -o pipefail
O=gpurun_out/r4t2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
B="python bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 50 --warmup 5"
timeout -k 10 200 $B > $O/c5.json 2>$O/c5.err && tail -1 $O/c5.json | cut -c1-200 &&
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && find $O/prof -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-160
