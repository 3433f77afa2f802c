set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/profile.sh r2 > gpurun_out/prof_r2.log 2>&1
W2XC_BENCH_ARGS="--precision bf16" bash tools/profile.sh r2_bf16 > gpurun_out/prof_r2_bf16.log 2>&1
W2XC_BENCH_ARGS="--precision fp16x2" bash tools/profile.sh r2_fp16x2 > gpurun_out/prof_r2_fp16x2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_profile_summary.py gpurun_out/prof_r2 gpurun_out/r2
python tools/make_profile_summary.py gpurun_out/prof_r2_bf16 gpurun_out/r2_bf16 bf16
python tools/make_profile_summary.py gpurun_out/prof_r2_fp16x2 gpurun_out/r2_fp16x2 fp16x2
cp gpurun_out/r2_roofline.json profiles/r2_roofline.json
python bench.py > gpurun_out/r2_bench_fp32.json 2> gpurun_out/bench_fp32.err; tail -c 600 gpurun_out/r2_bench_fp32.json
for p in bf16 fp16x2 bf16x2 bf16x3; do python bench.py --precision $p --no-cpu-baseline > gpurun_out/r2_bench_$p.json 2>/dev/null; done
python tools/run_configs.py > gpurun_out/r2_configs.json 2> gpurun_out/configs.err; tail -c 400 gpurun_out/r2_configs.json
