#!/bin/bash
# cycle shares (STOF_STAMPS builds) of several ablation builds in one session: bash tools/gpu_stamp_variants.sh s0 s1 ...
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for v in "$@"; do
  echo "== $v"
  STOF_LIB_PATH=stofnet_amd/libstof_$v.so timeout -k 10 200 python tools/read_stamps.py f16x3 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
done
