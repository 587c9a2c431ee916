#!/bin/bash
# Diagnostic / ablation build of the library: bash tools/build_variant.sh NAME -DFLAG [-DFLAG ...]  ->  stofnet_amd/libstof_NAME.so
# (only convstack.hip is rebuilt with the flags; use with STOF_LIB_PATH=stofnet_amd/libstof_NAME.so)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/stof_variant_$name
hipcc -O3 -std=c++17 -fPIC -fconstexpr-steps=100000000 --offload-arch=gfx950 -x hip -fno-slp-vectorize "$@" -c stofnet_amd/csrc/convstack.hip -o /tmp/stof_variant_$name/convstack.o
objs=""
for o in pack_weights shuffle_picker hilbert gradpeak neighbors train; do objs="$objs stofnet_amd/build/$o.o"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o stofnet_amd/libstof_$name.so /tmp/stof_variant_$name/convstack.o $objs
echo stofnet_amd/libstof_$name.so
