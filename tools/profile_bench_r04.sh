#!/bin/bash
# Round 4: the default bench line (C2 + extras.configs C3 / C4 / C5) + rocprofv3 evidence for the C2 run, the C5 step and the
# Hilbert / GradPeak kernels; outputs under gpurun_out/r04_*.  Stops at the first step that fails or times out.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
O=gpurun_out/r04
T="timeout -k 10 420"
$T python3 bench.py > ${O}_bench_default.json 2> ${O}_bench_default.err
$T python3 bench.py --config C5 --train-precision fp32 > ${O}_bench_c5_fp32.json 2> ${O}_bench_c5_fp32.err
$T python3 bench.py --precision fp32 --no-fp32-extra --no-extra-configs > ${O}_bench_c2_fp32.json 2> ${O}_bench_c2_fp32.err
$T python3 bench.py --config C5 --trainer autograd > ${O}_bench_c5_f16x3_autograd.json 2> ${O}_bench_c5_f16x3_autograd.err
echo benches done
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
passes() {      # passes NAME program args...: kernel stats + four counter passes of the same command, then the summary
  local name=$1; shift
  $T rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_${name}_kt -- "$@" > ${O}_${name}_kt.log 2>&1
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${O}_${name}_fetch -- "$@" > ${O}_${name}_fetch.log 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${O}_${name}_write -- "$@" > ${O}_${name}_write.log 2>&1
  $T rocprofv3 --kernel-trace --pmc $PMC1 --output-format csv -d ${O}_${name}_sq1 -- "$@" > ${O}_${name}_sq1.log 2>&1
  $T rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d ${O}_${name}_sq2 -- "$@" > ${O}_${name}_sq2.log 2>&1
  python3 tools/rocprof_summarize.py ${O}_${name}_pmc.json ${O}_${name}_kt ${O}_${name}_fetch ${O}_${name}_write ${O}_${name}_sq1 ${O}_${name}_sq2
  echo "$name passes done"
}
passes bench python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-extra --no-extra-configs
passes c5 python3 bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 10 --warmup 2
passes aux python3 tools/prof_hilbert.py
$T python3 tools/bench_small_batch.py > ${O}_small_batch.jsonl 2> ${O}_small_batch.err
echo profile_bench done
