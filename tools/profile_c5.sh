#!/bin/bash
# C5 (training step) bench lines + kernel stats; outputs under gpurun_out/r03_*
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
O=gpurun_out/r03
python3 bench.py --config C5 --train-precision f16x3 > ${O}_bench_c5_f16x3.json 2> ${O}_bench_c5_f16x3.err
python3 bench.py --config C5 --train-precision fp32 > ${O}_bench_c5_fp32.json 2> ${O}_bench_c5_fp32.err
python3 bench.py --config C5 --trainer autograd > ${O}_bench_c5_f16x3_autograd.json 2> ${O}_bench_c5_f16x3_autograd.err
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_c5_kt -- python3 bench.py --config C5 --no-cpu-baseline --steps 10 --warmup 2 > ${O}_c5_kt.log 2>&1
for f in f16x3 fp32 f16x3_autograd; do tail -1 ${O}_bench_c5_$f.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
