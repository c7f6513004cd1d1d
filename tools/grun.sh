#!/bin/bash
# Build container: rebuild every library first (a stale .so travels to the GPU box as it is), then gpurun.
# Usage: tools/grun.sh <timeout seconds> '<command>'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/crumble_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
make -s -C $R/tests/emu 2>&1 | grep -E "error|Error" && exit 1
make -s -C $R/oracle 2>&1 | grep -E "error|Error" && exit 1
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $t -- "$@"
