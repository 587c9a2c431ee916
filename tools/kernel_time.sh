#!/bin/bash
# kernel-only average durations (rocprofv3 --kernel-trace) of one tools/bench_aux.py case under a list of env settings
#   bash tools/kernel_time.sh CASE "VAR=a VAR=b ..."       (each word is exported for one run)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && export TMPDIR=/tmp
CASE=$1; shift
for setting in "$@"; do
  export $setting
  d=gpurun_out/kt_tmp; rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/bench_aux.py --case $CASE > /dev/null 2>&1
  python3 - "$setting" $d <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(list)
for f in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r['Kernel_Name'].replace('void (anonymous namespace)::','').split('(')[0][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v = sorted(v)
    print(f"{sys.argv[1]:28s} {k:72s} n={len(v):3d} median {v[len(v)//2]:8.2f} us  min {v[0]:8.2f}")
PY
  unset ${setting%%=*}
done
