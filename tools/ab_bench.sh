#!/bin/bash
# tools/ab_bench.sh <precision> ... -- A/B of two builds of the library inside ONE gpurun call (boxes differ by 2-4 %):
# ab/libw2xc_hip_old.so vs ab/libw2xc_hip_new.so are copied over lib/libw2xc_hip.so in turn and bench.py's resident leg is run
# AB_ROUNDS times per library, alternating.  The new library is left in place.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$REPO/waifu2x-converter-cpp_amd/lib/libw2xc_hip.so
for prec in "$@"; do
  for round in $(seq 1 ${AB_ROUNDS:-2}); do
    for which in old new; do
      cp $REPO/ab/libw2xc_hip_$which.so $LIB
      SWEEP_ARGS="--precision $prec" bash $REPO/tools/sweep_env.sh "AB=$which"
    done
  done
done
cp $REPO/ab/libw2xc_hip_new.so $LIB
