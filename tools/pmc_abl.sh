#!/bin/bash
# tools/pmc_abl.sh <tag> [env...] -- SQ counters (one pass) of the 128->128 layer bench; prints pipe-busy and clock
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_abl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python $REPO/tools/layer_bench.py --cin ${CIN:-128} --cout ${COUT:-128} --steps 3"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/a -o pmc --output-format csv -- $CMD > $OUT/a.log 2>&1
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/a/pmc_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "wino" not in k and "mfma" not in k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}; t = sum(dur[k]) / len(dur[k]) * 1e-9
    clk = m["GRBM_GUI_ACTIVE"] / 8 / t
    print("$TAG %-32s %.3f ms clock %.3f GHz  mfma_busy %.3f  wait_any %.3f  wait_inst %.3f  active %.3f (of wave cycles)" % (k[:32], t * 1e3, clk / 1e9,
          m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (clk * t), m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
