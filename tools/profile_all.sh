#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/<tag>_* is made of -- the default bench line, rocprofv3 kernel stats of the
# four single-GPU configurations, FETCH_SIZE / WRITE_SIZE of the step with the calibration kernels, SQ counters of
# k_columns2 on the chr20 set, the 300x panel and config 4 (separate --pmc passes, no trace domains; bench.py --profile:
# every launch in a trace is a timed step's).  Usage: tools/profile_all.sh <tag>; then, in the build container,
# python tools/summarize_profiles.py <tag>.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-rXX}
out=$R/gpurun_out
bash $R/tools/profile_round.sh $tag > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "chr20_9 --workload chr20 --preset 9" "panel --workload panel" "hifi_config4 --workload hifi"; do
  set -- $cfg; name=$1; shift
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-e2e "$@" > $out/${tag}_bench_$name.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$name -- python3 $R/bench.py --steps 3 --warmup 1 --profile "$@" > /dev/null 2>&1
done
# SQ counters of the column kernel: chr20 -5, the panel, config 4
for cfg in "chr20 --workload chr20" "panel --workload panel" "hifi --workload hifi"; do
  set -- $cfg; name=$1; shift
  n=0
  pass() { n=$((n+1)); rocprofv3 --pmc "$@" --output-format csv -d $out/${tag}_sq_${name}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --profile $BARGS > /dev/null 2>&1; }
  BARGS="$*"
  pass SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
  pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES
  pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
  if [ $name != chr20 ]; then      # (the chr20 traffic passes are profile_round.sh's)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch_$name -- python3 $R/bench.py --steps 1 --warmup 0 --profile $BARGS > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write_$name -- python3 $R/bench.py --steps 1 --warmup 0 --profile $BARGS > /dev/null 2>&1
  fi
done
# the device BGZF codec on a gigabyte of records (tools/gz_probe.py closes its handle before the interpreter exits: a destructor
# at exit hangs under the profiler)
python3 $R/tools/gz_probe.py 0.3 6 > $out/${tag}_codec_probe.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_codec -- python3 $R/tools/gz_probe.py 0.3 6 > /dev/null 2>&1
# the device pipeline (crumble-hip's default): its own clock over three runs beside the host pipeline's, then the kernel and copy
# trace of one run (the binary itself behind --), the inflate sweep
export TMPDIR=/tmp
PQ_KEEP=1 python3 $R/tools/pipe_quick.py 1 3 device1:CRUMBLE_BAM_LEVEL=1 device6: host1:CRUMBLE_HIP_PIPELINE=host,CRUMBLE_BAM_LEVEL=1 > $out/${tag}_pipeline.txt 2>&1
CRUMBLE_BAM_LEVEL=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/${tag}_prof_pipeline -- $R/crumble_amd/crumble-hip -5 -z /tmp/pq_in.bam /tmp/pq_out.bam > /dev/null 2>&1
python3 $R/tools/timeline.py $out/${tag}_prof_pipeline 0.8 0.4 > $out/${tag}_pipeline_timeline.txt 2>&1
rm -f /tmp/pq_in.bam /tmp/pq_out.bam
python3 $R/tools/inflate_sweep.py 0.01 0.05 0.2 0.6 > $out/${tag}_inflate_sweep.txt 2>&1
tail -c 600 $out/${tag}_bench.json
