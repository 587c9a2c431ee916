#!/bin/bash
# rocprofv3 evidence for the HBM-bound kernels: one rocprofv3 run per bench_aux case and per counter set
# (kernel trace; FETCH_SIZE; WRITE_SIZE -- separate --pmc passes, as MI355X_MICROARCH.md prescribes), then
# tools/aux_report.py folds them into gpurun_out/r03_aux_kernels.json (copied to profiles/ afterwards).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
OUT=gpurun_out/r03_aux
# PART=1 | 2 profiles the first / second half of the cases (one gpurun call each); the report is folded from both afterwards
PART=${PART:-0}
ALL=($(python3 tools/bench_aux.py --list))
H=$(( (${#ALL[@]} + 1) / 2 ))
if [ "$PART" = 1 ]; then CASES=("${ALL[@]:0:$H}"); python3 tools/bench_aux.py > ${OUT}_wall.jsonl 2> ${OUT}_wall.err
elif [ "$PART" = 2 ]; then CASES=("${ALL[@]:$H}")
else CASES=("${ALL[@]}"); python3 tools/bench_aux.py > ${OUT}_wall.jsonl 2> ${OUT}_wall.err; fi
for c in "${CASES[@]}"; do
  rm -rf ${OUT}_kt/$c ${OUT}_fetch/$c ${OUT}_write/$c
  rocprofv3 --kernel-trace --output-format csv -d ${OUT}_kt/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_fetch/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_write/$c -- python3 tools/bench_aux.py --case $c > /dev/null 2>&1
  echo "profiled $c"
done
if [ "$PART" != 1 ]; then python3 tools/aux_report.py ${OUT}; fi
