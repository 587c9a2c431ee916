# PMC passes over the [4096,2000] Hilbert kernel only (ONLY=2000 tools/prof_hilbert.py); summary JSON on stdout
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp ONLY=2000 REPS=6
OUT=${1:-gpurun_out/r2_hilct}
rm -rf ${OUT}_pmc*
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d ${OUT}_pmc1 -- python3 tools/prof_hilbert.py > ${OUT}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_IFETCH --output-format csv -d ${OUT}_pmc2 -- python3 tools/prof_hilbert.py > ${OUT}_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d ${OUT}_pmc3 -- python3 tools/prof_hilbert.py > ${OUT}_pmc3.log 2>&1
python3 tools/rocprof_summarize.py ${OUT}_summary.json ${OUT}_pmc1 ${OUT}_pmc2 ${OUT}_pmc3 > /dev/null 2>&1
python3 - ${OUT}_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def walk(o, depth=0):
    if isinstance(o, dict):
        for k, v in o.items():
            if isinstance(v, (dict, list)):
                print('  ' * depth + str(k)); walk(v, depth + 1)
            else:
                print('  ' * depth + f'{k}: {v}')
    elif isinstance(o, list):
        for v in o: walk(v, depth)
walk(d)
PY
