#!/bin/bash
# Round 4: the default bench line (C2 + extras.configs C3 / C4 / C5) + rocprofv3 evidence for the C2 run; outputs under gpurun_out/r04_*
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
O=gpurun_out/r04
python3 bench.py > ${O}_bench_default.json 2> ${O}_bench_default.err
python3 bench.py --config C5 --train-precision fp32 > ${O}_bench_c5_fp32.json 2> ${O}_bench_c5_fp32.err
python3 bench.py --precision fp32 --no-fp32-extra > ${O}_bench_c2_fp32.json 2> ${O}_bench_c2_fp32.err
echo benches done
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-extra --no-extra-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_bench_kt -- $B > ${O}_bench_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${O}_bench_fetch -- $B > ${O}_bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${O}_bench_write -- $B > ${O}_bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d ${O}_bench_sq1 -- $B > ${O}_bench_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d ${O}_bench_sq2 -- $B > ${O}_bench_sq2.log 2>&1
python3 tools/rocprof_summarize.py ${O}_bench_pmc.json ${O}_bench_kt ${O}_bench_fetch ${O}_bench_write ${O}_bench_sq1 ${O}_bench_sq2
python3 bench.py --config C5 --trainer autograd > ${O}_bench_c5_f16x3_autograd.json 2> ${O}_bench_c5_f16x3_autograd.err
python3 tools/report_precision.py > ${O}_precision.log 2>&1
python3 tools/bench_small_batch.py > ${O}_small_batch.jsonl 2> ${O}_small_batch.err
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_c5_kt -- python3 bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 10 --warmup 2 > ${O}_c5_kt.log 2>&1
echo profile_bench done
