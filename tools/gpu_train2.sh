#!/bin/bash
# training tests, then the C5 bench with and without the sparse SemiGlobalBlock backward
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_training.py -x -q -m gpu > gpurun_out/train_tests.log 2>&1
rc=$?; tail -4 gpurun_out/train_tests.log
[ $rc -ne 0 ] && exit $rc
for v in 1 0; do for dg in 1 0; do [ $v = 0 ] && [ $dg = 1 ] && continue; export STOF_TRAIN_SGB_SPARSE_DGRAD=$dg
  STOF_TRAIN_SGB_SPARSE=$v timeout -k 10 300 python3 bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sparse wgrad', $v, 'dgrad', $dg, d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
