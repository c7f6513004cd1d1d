#!/bin/bash
# CPU only: AddressSanitizer + UBSan over the whole host path AND the kernels' literal sources (compiled
# for the CPU through tests/emu/hip_emu.h), driven by the file -> file program on a small multi-contig BAM.
# (GPU ASan is not available on the pool; this is the sanitizer coverage the kernels get.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/crumble_asan}; mkdir -p $W; cd $W
SRC=$R/crumble_amd/csrc; EMU=$R/tests/emu
FL="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
gcc $FL -fPIC -c $SRC/params.c -o params.o
gcc $FL -fPIC -pthread -c $SRC/bamio.c -o bamio.o
g++ $FL -std=c++17 -fPIC -ffp-contract=off -DCRUMBLE_EMU -include $EMU/hip_emu.h -Wno-unknown-pragmas -x c++ -c $SRC/engine.hip -o engine.o
g++ $FL -std=c++17 -fPIC -c $EMU/hip_emu.cpp -o hip_emu.o
gcc $FL -c $SRC/cli.c -o cli.o
g++ $FL -o crumble-hip-asan cli.o engine.o hip_emu.o params.o bamio.o -lz -ldl -lpthread -lm
python3 - <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np
from crumble_amd import bamio, HostBatch
import synth
b0 = synth.generate(synth.config("ce", scale=0.5))
b1 = synth.generate(synth.config("hifi", scale=0.0002, read_len=3000, tid=1))
b2 = synth.generate(synth.config("panel", scale=0.0008, tid=2))
b0.quals = np.array([2, 11, 25, 37], np.uint8)[b0.quals % 4]      # a binned instrument's four values: CRUMBLE_HIP_CODED_QUALS sends them as codes
rng = np.random.default_rng(3); n, L = 30, 100
un = HostBatch(-1, np.full(n, -1), np.zeros(n), np.full(n, 4), np.arange(n + 1) * L, np.zeros(n + 1), np.zeros(0, np.uint32),
               rng.choice([1, 2, 4, 8, 15], n * L), rng.integers(2, 42, n * L))
bamio.write_batches("in.bam", [("a", 60000), ("b", 300000), ("c", 30000)], [b0, b1, b2, un])
open("k.bed", "w").write("a\t1000\t3000\nb\t500\t9000\n")
PY
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
# (the host pipeline's runs with the host codec: the emulated device deflate takes ten minutes per run under the sanitizers; the
# device codec is sanitized by the device-pipeline runs at the end, on a smaller file)
export CRUMBLE_HIP_CODEC=host CRUMBLE_HIP_PIPELINE=host
# (-j 3 / -j 3 -w with one emulated device named three times: the pthread pipeline, detach_group and run_windowed)
export CRUMBLE_HIP_DEVICES=0,0,0
for args in "-5" "-9 -R k.bed -b s.bed" "-1 -S -k 30-35" "-9 -y pbccs -T XS" "-3 -B -P 1.2 -C 0.01" "-5 -j 3 -b s3.bed" "-9 -j 3 -w -b s3w.bed -R k.bed"; do
  ./crumble-hip-asan $args in.bam out.bam > run.log 2>&1 || { echo "FAILED: $args"; tail -20 run.log; exit 1; }
  if grep -q "runtime error\|AddressSanitizer" run.log; then echo "SANITIZER REPORT: $args"; grep -m5 "runtime error\|AddressSanitizer" run.log; exit 1; fi
  echo "clean: $args"
done
# the qualities as 2-bit codes (crumble_pack_quals on the host's threads, k_p0_quals): contig a; b and c keep their bytes
for args in "-5" "-9 -j 3"; do
  CRUMBLE_HIP_CODED_QUALS=1 ./crumble-hip-asan $args in.bam out.bam > run.log 2>&1 || { echo "FAILED (coded qualities): $args"; tail -20 run.log; exit 1; }
  if grep -q "runtime error\|AddressSanitizer" run.log; then echo "SANITIZER REPORT (coded qualities): $args"; grep -m5 "runtime error\|AddressSanitizer" run.log; exit 1; fi
  echo "clean (coded qualities): $args"
done
# the device pipeline (crumble_stream: device inflate, record starts, engines fed from the records, device deflate; reader /
# writer threads of cli.c) on a smaller file -- cut into small feeds and drains, and whole
unset CRUMBLE_HIP_DEVICES CRUMBLE_HIP_CODEC
python3 - <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np
from crumble_amd import bamio, HostBatch
import synth
b0 = synth.generate(synth.config("ce", scale=0.08))
b1 = synth.generate(synth.config("hifi", scale=0.00005, read_len=3000, tid=1))
rng = np.random.default_rng(3); n, L = 10, 100
un = HostBatch(-1, np.full(n, -1), np.zeros(n), np.full(n, 4), np.arange(n + 1) * L, np.zeros(n + 1), np.zeros(0, np.uint32),
               rng.choice([1, 2, 4, 8, 15], n * L), rng.integers(2, 42, n * L))
bamio.write_batches("small.bam", [("a", 60000), ("b", 300000), ("c", 30000)], [b0, b1, un])
PY
for env in "CRUMBLE_HIP_DP_CHUNK=20000 CRUMBLE_HIP_DP_PIECE=10000" "CRUMBLE_HIP_ENGINES=1"; do
  for args in "-5" "-9 -R k.bed"; do
    env CRUMBLE_HIP_PIPELINE=device $env ./crumble-hip-asan -v $args small.bam out.bam > run.log 2>&1 || { echo "FAILED (device pipeline, $env): $args"; tail -20 run.log; exit 1; }
    if grep -q "runtime error\|AddressSanitizer" run.log; then echo "SANITIZER REPORT (device pipeline): $args"; grep -m5 "runtime error\|AddressSanitizer" run.log; exit 1; fi
    grep -q "pipeline: device" run.log || { echo "the device pipeline did not run: $args"; tail -5 run.log; exit 1; }
    echo "clean (device pipeline, $env): $args"
  done
done
