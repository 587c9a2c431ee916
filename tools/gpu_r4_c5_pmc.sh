#!/bin/bash
# r4: PMC counters of the C5 training step's kernels (separate passes, kernel-trace only)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04c5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_kt -- $B > ${O}_kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d ${O}_sq1 -- $B > ${O}_sq1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d ${O}_sq2 -- $B > ${O}_sq2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- $B > ${O}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${O}_write -- $B > ${O}_write.log 2>&1 &&
cd $GRAFT_REPO_ROOT && python3 tools/rocprof_summarize.py ${O}_pmc.json ${O}_kt ${O}_fetch ${O}_write ${O}_sq1 ${O}_sq2 && echo pmc done
