# PMC passes over tools/prof_hilbert.py (Hilbert envelope + GradPeak kernels); summaries via tools/rocprof_summarize.py
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
OUT=${1:-gpurun_out/r2_hil}
for T in 64 128 256 512; do STOF_HILBERT_THREADS=$T python tools/bench_aux.py hilbert 2>/dev/null | sed "s/^/T=$T /"; done > ${OUT}_threads.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d ${OUT}_pmc1 -- python3 tools/prof_hilbert.py > ${OUT}_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_IFETCH --output-format csv -d ${OUT}_pmc2 -- python3 tools/prof_hilbert.py > ${OUT}_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d ${OUT}_pmc3 -- python3 tools/prof_hilbert.py > ${OUT}_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_pmc4 -- python3 tools/prof_hilbert.py > ${OUT}_pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_pmc5 -- python3 tools/prof_hilbert.py > ${OUT}_pmc5.log 2>&1
tail -2 ${OUT}_pmc3.log
