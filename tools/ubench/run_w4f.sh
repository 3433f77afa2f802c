#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 160"
for s in "64 64" "64 128" "128 64" "128 128"; do
  $T ./wino4_timing $s 50 70 0 0 0 0 0 1 | tail -1
  $T ./wino4_timing $s 61 67 0 2 3 0 0 1 | tail -1
done
$T ./wino4_timing 32 64 33 300 0 3 5 2 1 1 | tail -1
$T ./wino4_timing 32 128 50 70 0 0 0 0 1 1 | tail -1
$T ./wino4_timing 128 128 2160 3840 0 0 0 0 0 1 | tail -3
$T ./wino4_timing 128 128 2160 3840 0 0 0 0 0 0 | tail -2
$T ./wino4_timing 128 128 2160 3840 0 0 0 0 0 1 | tail -3
