#!/bin/bash
# r4: body sweep variants -- parity, stamps, bench line (no extras), training tests
set -o pipefail
O=gpurun_out/r4b2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
STOF_LIB_PATH=stofnet_amd/libstof_stamps.so timeout -k 10 300 python tools/read_stamps.py > $O/stamps.txt 2>&1 && cat $O/stamps.txt | cut -c1-100 &&
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp32-extra --no-extra-configs > $O/c2.json 2> $O/c2.err && python - $O/c2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('C2', d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
PY
timeout -k 10 200 python bench.py --config C5 --no-cpu-baseline --no-extra-configs --steps 50 --warmup 5 > $O/c5.json 2>$O/c5.err && tail -1 $O/c5.json | cut -c1-200
