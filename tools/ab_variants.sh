#!/bin/bash
# Runs ON THE GPU BOX: quick bench lines for several library variants built by tools/build_variant.sh.
# Usage: tools/ab_variants.sh <tag> [<tag> ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for tag in "$@"; do
  echo "== $tag"
  CRUMBLE_HIP_LIB=$R/build/libcrumble_$tag.so bash $R/tools/quick_bench.sh
done
