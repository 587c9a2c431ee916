#!/bin/bash
# timing experiment on a variant library (results of the variant are NOT correct): kernel stats of the C5 step only
set -o pipefail
O=gpurun_out/r4wexp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  STOF_LIB_PATH=$GRAFT_REPO_ROOT/stofnet_amd/libstof_$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/prof_$v.log; exit 1; }
  echo "== $v"; grep -E "wgrad_split_async|wgrad_f16x3_batch" $GRAFT_REPO_ROOT/$O/prof_$v/c5_kernel_stats.csv | cut -d, -f1-7 | cut -c1-200
done
