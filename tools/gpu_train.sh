#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py -m gpu -x -q -k "training or train or conv_kernels or autograd or semi_global or other_semi or main_entry or ddp or loss or adamw" > gpurun_out/r03e_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r03e_pytest.log
python tools/time_train_parts.py f16x3 2>/dev/null
for tr in fused autograd; do python bench.py --config C5 --train-precision f16x3 --trainer $tr --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['trainer'], d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
