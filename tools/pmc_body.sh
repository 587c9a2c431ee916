#!/bin/bash
# instruction-mix / stall counters of the forward kernels (two PMC passes + kernel trace), C2 shape, 5 steps
#   bash tools/pmc_body.sh TAG        -> gpurun_out/pmc_body_TAG.json
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
T=${1:-x}
O=gpurun_out/pmc_body_$T
rm -rf ${O}_kt ${O}_sq1 ${O}_sq2 ${O}_sq3
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-extra"
rocprofv3 --kernel-trace --output-format csv -d ${O}_kt -- $B > ${O}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d ${O}_sq1 -- $B > ${O}_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d ${O}_sq2 -- $B > ${O}_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT --output-format csv -d ${O}_sq3 -- $B > ${O}_sq3.log 2>&1
python3 tools/rocprof_summarize.py ${O}.json ${O}_kt ${O}_sq1 ${O}_sq2 ${O}_sq3 > /dev/null
python3 - ${O}.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,g in d['groups'].items():
    if 'body_sweep' in k or 'sgb_contract' in k:
        print(k, g.get('mean_us'))
        for c,v in sorted(g.get('counters',{}).items()): print('   ',c,v)
PY
