#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q > gpurun_out/r03e_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r03e_pytest.log
python tools/time_train_parts.py f16x3 2>/dev/null
STOF_TRAIN_SWEEP=0 python tools/time_train_parts.py f16x3 2>/dev/null
for tr in fused autograd; do python bench.py --config C5 --train-precision f16x3 --trainer $tr --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['trainer'], d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
