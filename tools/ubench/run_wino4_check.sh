#!/bin/bash
# gpurun -- 'bash tools/ubench/run_wino4_check.sh [full]' : conv3x3_wino4 alone (tools/ubench/wino4_timing.hip) -- every shape and output mode against a
# double-precision direct sum on small and ragged planes; `full`: the 2160x3840 layer times too.  Build wino4_timing first (its header has the command).
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 120"
echo "== correctness, small planes =="
for s in "32 64" "32 128" "64 64" "64 128" "128 64" "128 128"; do
  $T ./wino4_timing $s 50 70 0 | tail -1
  $T ./wino4_timing $s 50 70 1 | tail -1
done
$T ./wino4_timing 128 128 33 37 | tail -1
$T ./wino4_timing 128 128 100 300 | tail -1
$T ./wino4_timing 64 64 40 600 | tail -1
$T ./wino4_timing 64 64 40 600 1 | tail -1
$T ./wino4_timing 128 128 61 67 0 1 0 | tail -1
$T ./wino4_timing 128 128 61 67 0 2 3 | tail -1
$T ./wino4_timing 128 128 61 67 0 3 5 2 | tail -1
$T ./wino4_timing 32 64 4 4 | tail -1
$T ./wino4_timing 32 64 1 1 | tail -1
echo "== 32 planes in, NHWC =="
for s in "32 64" "32 128"; do
  $T ./wino4_timing $s 50 70 0 0 0 0 1 | tail -1
  $T ./wino4_timing $s 61 67 1 2 3 0 1 | tail -1
  $T ./wino4_timing $s 33 300 0 3 5 2 1 | tail -1
done
echo "== fused last layer =="
for s in "64 64" "64 128" "128 64" "128 128"; do
  $T ./wino4_timing $s 50 70 0 0 0 0 0 1 | tail -1
  $T ./wino4_timing $s 61 67 0 2 3 0 0 1 | tail -1
done
$T ./wino4_timing 32 64 33 300 0 3 5 2 1 1 | tail -1
[ "$1" = "full" ] || exit 0
echo "== full frame =="
for s in "128 128" "64 128" "64 64" "32 64"; do
  $T ./wino4_timing $s | tail -3
done
$T ./wino4_timing 128 128 2160 3840 0 0 0 0 0 1 | tail -3
