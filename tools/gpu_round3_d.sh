#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03d_pytest.log
for c in C2 C3 C4; do
python bench.py --config $c $( [ $c = C4 ] && echo "--steps 5 --warmup 1" ) > gpurun_out/r03d_bench_$c.json 2> gpurun_out/r03d_bench_$c.err
python - gpurun_out/r03d_bench_$c.json $c <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity_on_sample'))
PY
done
