#!/bin/bash
# Diagnostic / ablation build of the library: bash tools/build_variant.sh NAME -DFLAG [-DFLAG ...]  ->  stofnet_amd/libstof_NAME.so
# (only one source is rebuilt with the flags -- SRC=convstack (default) | gradpeak | hilbert | ...; use with STOF_LIB_PATH=stofnet_amd/libstof_NAME.so)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p /tmp/stof_variant_$name
src=${SRC:-convstack}
extra=""; [ "$src" = convstack ] && extra="-fno-slp-vectorize"
hipcc -O3 -std=c++17 -fPIC -fconstexpr-steps=100000000 --offload-arch=gfx950 -x hip $extra "$@" -c stofnet_amd/csrc/$src.hip -o /tmp/stof_variant_$name/$src.o
objs=""
for o in pack_weights convstack shuffle_picker hilbert gradpeak neighbors train; do [ "$o" = "$src" ] || objs="$objs stofnet_amd/build/$o.o"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o stofnet_amd/libstof_$name.so /tmp/stof_variant_$name/$src.o $objs
echo stofnet_amd/libstof_$name.so
