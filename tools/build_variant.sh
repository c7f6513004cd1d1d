#!/bin/bash
# Builds build/libcrumble_<tag>.so from the working tree with extra -D flags (A/B tuning on the GPU box:
# CRUMBLE_HIP_LIB=build/libcrumble_<tag>.so python bench.py ...).  Usage: tools/build_variant.sh <tag> [-DKCOL_PERM=0 ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
mkdir -p $R/build
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result "$@" \
    -c $R/crumble_amd/csrc/engine.hip -o $R/build/engine_$tag.o
/opt/rocm/bin/hipcc -shared -fPIC -Wl,-Bsymbolic --offload-arch=gfx950 -o $R/build/libcrumble_$tag.so \
    $R/crumble_amd/csrc/params.o $R/build/engine_$tag.o -lm -lpthread
echo built $R/build/libcrumble_$tag.so
