#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2..: PMC counters, each in its
# own run with --kernel-trace only (never combined with sys/hip/hsa tracing).
# Results land in gpurun_out/prof_<tag>/ ; copy the summaries to profiles/ afterwards.
TAG=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-host $W2XC_BENCH_ARGS"   # e.g. W2XC_BENCH_ARGS="--precision bf16x3"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES -d $OUT/pmc_sq -o pmc --output-format csv -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_lds -o pmc --output-format csv -- $BENCH > $OUT/pmc_lds.log 2>&1
# issue-time accounting (does the matrix pipe co-execute with the VALU?) and the memory / LDS pipelines' full-queue stalls
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_issue -o pmc --output-format csv -- $BENCH > $OUT/pmc_issue.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $OUT/pmc_fifo -o pmc --output-format csv -- $BENCH > $OUT/pmc_fifo.log 2>&1
find $OUT -name "*.csv" | head -40
tail -3 $OUT/*.log
