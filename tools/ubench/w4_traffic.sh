#!/bin/bash
# Where do the L2 misses of conv3x3_wino4 <128,128> come from?  (prepared at the end of round 3: profiles/r3_sweeps.log block 22)
# Build here (no GPU needed), run on the GPU box:
#   bash tools/ubench/w4_traffic.sh build        # cross-compiles the harness variants into tools/ubench/w4v_*
#   gpurun --timeout 420 -- 'bash tools/ubench/w4_traffic.sh run'
# Every GPU command has its own SHORT timeout and ONE counter per rocprofv3 pass: in round 3 a pass with FETCH_SIZE and WRITE_SIZE together aborted
# (signal 6) and four such passes under `timeout 200` burnt ten GPU-minutes.
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
CSRC=$HERE/../../waifu2x-converter-cpp_amd/csrc
VARIANTS="base:-DW4_ABL=0 nou:-DW4_ABL=16 noraw:-DW4_ABL=32 nost:-DW4_ABL=64"
if [ "${1:-}" = build ]; then
    for v in $VARIANTS; do
        n=${v%%:*}; f=${v#*:}
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DW4_TIMING $f -I$CSRC $HERE/wino4_timing.hip -o $HERE/w4v_$n &
    done
    wait; ls -la $HERE/w4v_*; exit 0
fi
cd /tmp && export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-$HERE/../..}/gpurun_out/w4v; mkdir -p $O
for v in $VARIANTS; do
    n=${v%%:*}
    echo "== $n"; timeout 40 $HERE/w4v_$n 128 128 | head -3
done > $O/timing.txt 2>&1
for v in $VARIANTS; do
    n=${v%%:*}
    for ctr in FETCH_SIZE WRITE_SIZE TCC_MISS_sum TCC_HIT_sum; do
        timeout 45 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${n}_$ctr -o p -- $HERE/w4v_$n 128 128 > $O/${n}_$ctr.log 2>&1 || echo "$n $ctr: rc $?" >> $O/timing.txt
        f=$(find $O/${n}_$ctr -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && python3 - "$f" "$n" "$ctr" >> $O/pmc.txt <<'PY'
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'wino4' in r['Kernel_Name']]
print(sys.argv[2], sys.argv[3], ' '.join('%.5g' % x for x in v))
PY
        rm -rf $O/${n}_$ctr
    done
done
cat $O/timing.txt $O/pmc.txt
