#!/bin/bash
# rocprofv3 PMC passes over one harness binary: bash tools/ubench/pmc_w4p.sh <binary> [args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$1; shift
OUT=$R/gpurun_out/pmc_$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { timeout 200 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o pmc --output-format csv -- $R/tools/ubench/$B "$@" > $OUT/$1.log 2>&1; }
P=$1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/sq -o pmc --output-format csv -- $R/tools/ubench/$B 128 128 > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $OUT/lds -o pmc --output-format csv -- $R/tools/ubench/$B 128 128 > $OUT/lds.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "lds"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino4p" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-28s n=%d last=%.4g" % (k, len(v), v[-1]))
PY
