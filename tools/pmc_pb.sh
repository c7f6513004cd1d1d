R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $R/gpurun_out/pmc9 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --preset 9 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/pmc9/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_pb_reads" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: int(v) for k, v in sorted(tot.items())})
PY
