#!/bin/bash
# GradPeak parity tests, then kernel-only timings of the GradPeak cases of tools/bench_aux.py
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gradpeak or pala or long_rows" > gpurun_out/gp_tests.log 2>&1
rc=$?; tail -5 gpurun_out/gp_tests.log
[ $rc -ne 0 ] && exit $rc
for c in gradpeak_th1em3_2048x2000_rf10 gradpeak_th1em3_4096x2000_rf10 gradpeak_default_th_2048x2000_rf10 gradpeak_default_th_4096x2000_rf10 gradpeak_unfused_4096x4000_rf20_th1em3 gradpeak_long_512x30720_rf20_th1em4 ${EXTRA_CASES:-}; do
  echo "== $c"
  bash tools/kernel_time.sh $c X=1 2>&1 | sed 's/^X=1 *//' | tee -a gpurun_out/gp_times.log
done
