#!/bin/bash
# rocprofv3 evidence for the HBM-bound kernels: one rocprofv3 run per bench_aux case and per counter set
# (kernel trace; FETCH_SIZE; WRITE_SIZE -- separate --pmc passes, as MI355X_MICROARCH.md prescribes), then
# tools/aux_report.py folds them into gpurun_out/r03_aux_kernels.json (copied to profiles/ afterwards).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
OUT=gpurun_out/r03_aux
rm -rf ${OUT}_kt ${OUT}_fetch ${OUT}_write
python3 tools/bench_aux.py > ${OUT}_wall.jsonl 2> ${OUT}_wall.err
for c in $(python3 tools/bench_aux.py --list); do
  rocprofv3 --kernel-trace --output-format csv -d ${OUT}_kt/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_fetch/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_write/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  echo "profiled $c"
done
python3 tools/aux_report.py ${OUT}
