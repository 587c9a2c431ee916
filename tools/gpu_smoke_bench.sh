#!/bin/bash
# what the driver runs at round end, minus the tests: smoke(), then the default bench line
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 500 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_default.json').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,c in d['extras']['configs'].items():
    print(k, c.get('value'), c.get('ms_per_step'), c.get('roofline',{}).get('frac'), c.get('roofline',{}).get('algorithmic_frac'), c.get('child_wall_s'), c.get('error'))
PY
