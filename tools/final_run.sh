set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile.sh r5 > gpurun_out/prof_r5.log 2>&1
W2XC_BENCH_ARGS="--precision bf16" bash tools/profile.sh r5_bf16 > gpurun_out/prof_r5_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_profile_summary.py gpurun_out/prof_r5 gpurun_out/r5 fp32 2>&1 | tail -3
python tools/make_profile_summary.py gpurun_out/prof_r5_bf16 gpurun_out/r5_bf16 bf16 2>&1 | tail -3
cp gpurun_out/r5_roofline.json profiles/r5_roofline.json
cp gpurun_out/r5_bf16_roofline.json profiles/r5_bf16_roofline.json
python bench.py > gpurun_out/r5_bench_fp32.json 2> gpurun_out/bench_fp32.err; tail -c 900 gpurun_out/r5_bench_fp32.json
python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/r5_bench_bf16.json 2> gpurun_out/bench_bf16.err; tail -c 700 gpurun_out/r5_bench_bf16.json
python tools/run_configs.py > gpurun_out/configs.log 2> gpurun_out/configs.err; cp gpurun_out/configs.json gpurun_out/r5_configs.json; tail -c 600 gpurun_out/configs.log
