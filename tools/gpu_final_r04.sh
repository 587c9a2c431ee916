#!/bin/bash
# round-4 final: the whole GPU suite, then the default bench line and its profiles (tools/profile_bench_r04.sh)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_tests.log 2>&1 || { tail -40 gpurun_out/r04_final_tests.log; exit 1; }
tail -3 gpurun_out/r04_final_tests.log
