#!/bin/bash
# first GPU pass of round 3: the whole -m gpu suite, the bench line, GradPeak exactness, N-rank rehearsal
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/r03a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03a_pytest.log
tail -3 gpurun_out/r03a_pytest.log
python bench.py > gpurun_out/r03a_bench_c2.json 2> gpurun_out/r03a_bench_c2.err && tail -c 600 gpurun_out/r03a_bench_c2.json
python tools/gradpeak_exactness.py > gpurun_out/r03a_exactness.log 2>&1; tail -8 gpurun_out/r03a_exactness.log
bash tools/rehearse_ranks.sh > gpurun_out/r03a_rehearse.log 2>&1; tail -2 gpurun_out/r03a_rehearse.log
