#!/bin/bash
# correctness + speed of the two-pass tile-major body (r4) against the r3 chunk-major kernel (STOF_BODY_P2=0), same box, same session
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${KEXPR:-forward or argmax or sgb or config or onsets or auto or short or small}" > gpurun_out/r04_p2_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r04_p2_pytest.log
[ $rc -ne 0 ] && [ -z "${BENCH_ANYWAY:-}" ] && exit $rc
for v in ${ORDER:-1 0 1}; do
STOF_BODY_P2=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra > gpurun_out/r04_bench_p2_$v.json 2> gpurun_out/r04_bench_p2_$v.err || { echo "bench P2=$v failed"; tail -5 gpurun_out/r04_bench_p2_$v.err; exit 1; }
python - gpurun_out/r04_bench_p2_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('BODY_P2='+sys.argv[2], d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
PY
done
