#!/bin/bash
# same-box A/B of ablation builds of the library: bash tools/gpu_variants.sh base v1 v2 ...   (base = stofnet_amd/libstofnet_amd.so)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for v in "$@"; do
  lib=stofnet_amd/libstof_$v.so; [ "$v" = base ] && lib=stofnet_amd/libstofnet_amd.so
  STOF_LIB_PATH=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra ${BENCH_ARGS:-} > gpurun_out/variant_$v.json 2> gpurun_out/variant_$v.err || { echo "bench $v failed"; tail -3 gpurun_out/variant_$v.err; continue; }
  python - gpurun_out/variant_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2].ljust(8), d['value'], d['ms_per_step'], d['kernels_ms']['body_sweep'], d['roofline']['frac'])
PY
done
