#!/bin/bash
# tools/pmc_layer.sh <precision> -- SQ counters of one 128->128 layer bench (tuning aid; run through gpurun)
P=${1:-3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_layer_$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${W2XC_PMC_CMD:-"python $REPO/tools/layer_bench.py --precision $P --steps 3"}
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o pmc --output-format csv -- $CMD > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $OUT/b -o pmc --output-format csv -- $CMD > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/c -o pmc --output-format csv -- $CMD > $OUT/c.log 2>&1
python - <<PY
import csv,glob,collections
for sub in "abc":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in agg.items():
            if "conv3x3" in k: print(k, {a:round(b) for a,b in v.items()})
PY
