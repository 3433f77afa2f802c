#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
echo "== old kernel (round 3) =="; timeout 120 ./wino4_r3_timing 128 128 | head -3
bash ./run_w4p2.sh "$@"
echo "== old kernel (round 3) =="; timeout 120 ./wino4_r3_timing 128 128 | head -3
