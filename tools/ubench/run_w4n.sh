#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 120"
for s in "32 64" "32 128"; do
  $T ./wino4_timing $s 50 70 0 0 0 0 1 | tail -1
  $T ./wino4_timing $s 50 70 1 0 0 0 1 | tail -1
  $T ./wino4_timing $s 61 67 0 2 3 0 1 | tail -1
  $T ./wino4_timing $s 33 300 0 3 5 2 1 | tail -1
done
$T ./wino4_timing 32 64 2160 3840 0 0 0 0 1 | tail -2
$T ./wino4_timing 32 64 2160 3840 0 0 0 0 0 | tail -2
