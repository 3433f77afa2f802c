#!/bin/bash
# gpurun -- 'bash tools/ubench/run_w4p.sh' : conv3x3_wino4p alone -- correctness on small and ragged planes, then time against the round-3 kernel
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 120"
echo "== correctness, small planes =="
for s in "32 64" "32 128" "64 64" "64 128" "128 64" "128 128"; do
  $T ./wino4p_timing $s 50 70 0 | tail -1
  $T ./wino4p_timing $s 50 70 1 | tail -1
done
$T ./wino4p_timing 128 128 33 37 | tail -1
$T ./wino4p_timing 128 128 100 300 | tail -1
$T ./wino4p_timing 64 64 40 600 | tail -1
$T ./wino4p_timing 64 64 40 600 1 | tail -1
$T ./wino4p_timing 128 128 61 67 0 1 0 | tail -1
$T ./wino4p_timing 128 128 61 67 0 2 3 | tail -1
$T ./wino4p_timing 128 128 61 67 0 3 5 2 | tail -1
$T ./wino4p_timing 32 64 4 4 | tail -1
$T ./wino4p_timing 32 64 1 1 | tail -1
exit 0
echo "== full frame =="
for s in "128 128" "64 128" "64 64" "32 64"; do
  $T ./wino4p_timing $s | tail -3
  $T ./wino4_timing $s | head -3
done
$T ./wino4p_timing 128 128 2160 3840 1 | tail -3
echo "== stamps =="
$T ./wino4p_timing_st 128 128 | tail -40
echo "== ablations (128->128) =="
for a in 2 3 16 32 64; do echo "W4_ABL=$a"; $T ./wino4p_abl$a 128 128 | tail -2; done
