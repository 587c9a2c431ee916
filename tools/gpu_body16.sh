#!/bin/bash
# correctness + speed of the 16x16x32 body against the 32x32x16 form (STOF_BODY16=0), same box, same session
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or argmax or sgb or config or onsets or auto or short or small" > gpurun_out/r03c_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03c_pytest.log
for v in ${ORDER:-1 0 1}; do
STOF_BODY16=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra > gpurun_out/r03c_bench_body16_$v.json 2> gpurun_out/r03c_bench_$v.err
python - gpurun_out/r03c_bench_body16_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('BODY16='+sys.argv[2], d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
PY
done
STOF_LIB_PATH=stofnet_amd/libstof_stamps2.so timeout -k 10 200 python tools/read_stamps.py f16x3 2>/dev/null
