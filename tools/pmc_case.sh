#!/bin/bash
# Instruction-mix counters of one tools/bench_aux.py case:  bash tools/pmc_case.sh CASE [TAG]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
CASE=$1; TAG=${2:-x}
OUT=gpurun_out/pmc_${CASE}_${TAG}
rm -rf ${OUT}_1 ${OUT}_2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d ${OUT}_1 -- python3 tools/bench_aux.py --case $CASE > /dev/null 2>&1
python3 tools/rocprof_summarize.py ${OUT}.json ${OUT}_1 > /dev/null 2>&1
python3 - ${OUT}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, g in d['groups'].items():
    if g['calls'] < 5: continue
    c = g['counters']
    print(k[:60], 'calls', g['calls'], 'us', g.get('mean_us'), {n: round(v / 1e6, 3) for n, v in c.items()})
PY
