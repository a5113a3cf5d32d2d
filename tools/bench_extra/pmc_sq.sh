#!/bin/bash
# SQ counters of the count kernels on configs[1] (two passes of 8 / 5 counters); per-kernel averages to stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-pmc_sq}
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 bench.py --workload cfg1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- python3 bench.py --workload cfg1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/b.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][:40]
            if not name.replace("void ", "").startswith("k_"): continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in sorted(acc.items()):
        print(name, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
