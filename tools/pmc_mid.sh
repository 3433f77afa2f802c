#!/bin/bash
# tools/pmc_mid.sh <cin> <cout> <tag> [env assignments...] -- SQ / LDS / TCC counters of ONE fp32 mid layer (tools/layer_bench.py) under
# rocprofv3, each counter group in its own pass (--kernel-trace only).  Tuning aid; run through gpurun.
CIN=$1; COUT=$2; TAG=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_mid_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python $REPO/tools/layer_bench.py --cin $CIN --cout $COUT --steps 3"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/a -o pmc --output-format csv -- $CMD > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/b -o pmc --output-format csv -- $CMD > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/c -o pmc --output-format csv -- $CMD > $OUT/c.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/d -o pmc --output-format csv -- $CMD > $OUT/d.log 2>&1
grep -h "TFLOP" $OUT/a.log | tail -1
python $REPO/tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | sort) | grep -v "first\|last" | tee $OUT/summary.txt
