#!/bin/bash
# tools/sweep_env.sh "<ENV=val ...>" ... -- bench.py (resident leg only) under several tuning-env settings; one line per setting
# with the per-layer milliseconds.  Run through gpurun; results also in gpurun_out/sweep.log.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
for cfg in "$@"; do
  out=$(env $cfg python $REPO/bench.py --steps ${SWEEP_STEPS:-5} --warmup 2 --no-host --no-extras --no-cpu-baseline ${SWEEP_ARGS} 2>/dev/null | tail -1)
  python - "$cfg" <<PY "$out" | tee -a $REPO/gpurun_out/sweep.log
import json,sys
cfg=sys.argv[1]
try:
    j=json.loads(sys.argv[2])
    print("%-44s %7.3f ms/step %6.2f Mpix/s | %s" % (cfg, j["ms_per_step"], j["value"], " ".join("L%d %.3f" % (l["layer"], l["ms"]) for l in j["layers"])))
except Exception as e:
    print(cfg, "FAILED", repr(e), sys.argv[2][:200])
PY
done
