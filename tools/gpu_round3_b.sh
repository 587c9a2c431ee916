#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s -k "training or autograd or semi_global or other_semi or pala or main_entry or sgb or short_rows" > gpurun_out/r03b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03b_pytest.log
tail -5 gpurun_out/r03b_pytest.log
for tp in fp32 f16x3; do for tr in fused autograd; do
python bench.py --config C5 --train-precision $tp --trainer $tr --no-cpu-baseline > gpurun_out/r03b_c5_${tp}_${tr}.json 2> gpurun_out/r03b_c5_${tp}_${tr}.err; python - gpurun_out/r03b_c5_${tp}_${tr}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['config']['precision'],d['config']['trainer'],d['value'],d['ms_per_step'],d['roofline']['frac'])
PY
done; done
