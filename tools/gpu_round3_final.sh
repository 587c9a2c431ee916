#!/bin/bash
# end-of-round evidence: the whole -m gpu suite, smoke, C5 lines + kernel stats, N-rank rehearsal on one GPU
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03f_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03f_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --config C5 > gpurun_out/r03_bench_c5_f16x3.json 2> gpurun_out/c5.err
python3 bench.py --config C5 --trainer autograd > gpurun_out/r03_bench_c5_f16x3_autograd.json 2>> gpurun_out/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_c5_kt4 -- python3 bench.py --config C5 --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/r03_c5_kt4.log 2>&1
bash tools/rehearse_ranks.sh > gpurun_out/r03f_rehearse.log 2>&1; tail -1 gpurun_out/r03f_rehearse.log
python3 bench.py > gpurun_out/r03f_bench_c2.json 2> gpurun_out/r03f_bench_c2.err; tail -c 300 gpurun_out/r03f_bench_c2.json
