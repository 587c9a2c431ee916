#!/bin/bash
# A/B of library variants on one box: bash tools/gpu_ab.sh "libA.so libB.so ..." (paths under stofnet_amd/; "default" = the product build)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for rep in 1 2; do for v in $1; do
if [ "$v" = default ]; then unset STOF_LIB_PATH; else export STOF_LIB_PATH=stofnet_amd/$v; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra ${BENCH_ARGS:-} > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
python - gpurun_out/ab_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['extras'].get('parity_check',''))
PY
done; done
