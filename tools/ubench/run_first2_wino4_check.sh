#!/bin/bash
# gpurun -- 'bash tools/ubench/run_first2_wino4_check.sh [full]' : conv3x3_first2_wino4 alone (tools/ubench/first2_wino4_timing.hip) against a double-precision evaluation of
# both layers on small and ragged planes, block phases, folded padding (off < 0) and the folded nearest-2x; `full`: the 2160x3840 time too.  Build first2_wino4_timing first.
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 120"
echo "== correctness, small planes =="
$T ./first2_wino4_timing 50 70 | tail -1
$T ./first2_wino4_timing 61 67 2 | tail -1
$T ./first2_wino4_timing 33 300 3 | tail -1
$T ./first2_wino4_timing 1 1 | tail -1
$T ./first2_wino4_timing 4 4 1 | tail -1
$T ./first2_wino4_timing 100 300 0 -2 | tail -1
$T ./first2_wino4_timing 40 600 1 -7 | tail -1
$T ./first2_wino4_timing 64 96 0 -2 1 | tail -1
$T ./first2_wino4_timing 51 77 3 -7 1 | tail -1
[ "$1" = "full" ] || exit 0
echo "== full frame =="
$T ./first2_wino4_timing | tail -3
$T ./first2_wino4_timing 2160 3840 0 -7 1 | tail -2
