#!/bin/bash
# tools/final_run.sh -- everything a round's closing measurement needs, in ONE gpurun call (outputs under gpurun_out/, summaries copied to profiles/ by the caller):
#   gpurun --timeout 2400 -- 'bash tools/final_run.sh r6'
TAG=${1:-r6}
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
# race screens: the host pipeline (300 random conversions vs the resident path) and the concurrent-caller program under ThreadSanitizer (engine host code instrumented;
# the HIP / HSA runtimes are not: tools/tsan.supp names them and nothing else)
python tools/stress_host_pipeline.py --iters 300 > gpurun_out/${TAG}_stress_host_pipeline.log 2>&1; tail -1 gpurun_out/${TAG}_stress_host_pipeline.log
if [ -x waifu2x-converter-cpp_amd/lib/tsan/thread_stress ]; then
  (cd waifu2x-converter-cpp_amd/lib/tsan && TSAN_OPTIONS="suppressions=$GRAFT_REPO_ROOT/tools/tsan.supp" timeout 900 ./thread_stress 4 3 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tsan.log 2>&1; echo "thread_stress under tsan: exit code $?" >> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tsan.log)
  tail -3 gpurun_out/${TAG}_tsan.log
fi
bash tools/profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
W2XC_BENCH_ARGS="--precision bf16" bash tools/profile.sh ${TAG}_bf16 > gpurun_out/prof_${TAG}_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_profile_summary.py gpurun_out/prof_$TAG gpurun_out/$TAG fp32 2>&1 | tail -8
python tools/make_profile_summary.py gpurun_out/prof_${TAG}_bf16 gpurun_out/${TAG}_bf16 bf16 2>&1 | tail -3
cp gpurun_out/${TAG}_roofline.json profiles/${TAG}_roofline.json
cp gpurun_out/${TAG}_bf16_roofline.json profiles/${TAG}_bf16_roofline.json
python bench.py > gpurun_out/${TAG}_bench_fp32.json 2> gpurun_out/bench_fp32.err; tail -c 1200 gpurun_out/${TAG}_bench_fp32.json
python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/bench_bf16.err; tail -c 400 gpurun_out/${TAG}_bench_bf16.json
python tools/host_ab.py > gpurun_out/${TAG}_host_ab.log 2>&1; python tools/host_ab.py --pinned >> gpurun_out/${TAG}_host_ab.log 2>&1; grep -v amdgpu.ids gpurun_out/${TAG}_host_ab.log
python tools/fusion_ab.py --rounds 3 > gpurun_out/${TAG}_fusion_ab.log 2>&1; grep round gpurun_out/${TAG}_fusion_ab.log
python tools/run_configs.py > gpurun_out/configs.log 2> gpurun_out/configs.err; cp gpurun_out/configs.json gpurun_out/${TAG}_configs.json; tail -c 600 gpurun_out/configs.log
python tools/shard_projection.py > gpurun_out/${TAG}_shard_projection.log 2>&1; tail -3 gpurun_out/${TAG}_shard_projection.log
