#!/bin/bash
# weight-gradient kernel forms on one box: tests, then kernel stats of the C5 step per form (STOF_TRAIN_WGRAD_ASYNC = 1 | 2 | 0)
set -o pipefail
O=gpurun_out/r4wf; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in ${FORMS:-1 2 0}; do
  export STOF_TRAIN_WGRAD_ASYNC=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/prof_$v.log; exit 1; }
  echo "== form $v"; grep -E "wgrad_split_async|wgrad_f16x3_batch" $GRAFT_REPO_ROOT/$O/prof_$v/c5_kernel_stats.csv | cut -d, -f1-7 | cut -c1-200
done
