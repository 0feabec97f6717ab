#!/bin/bash
# usage (on the GPU box): bash tools/round_measurements.sh   -> gpurun_out/r03_bench_*.json, prof_* summaries
# The round's committed measurements: default bench line (with the CPU baseline), rocprofv3 summaries of the C2 / C4 / C5 steps,
# C1, forced-DP, float16 / float32 and C5-in-float16 lines.  Copy what is to be judged into profiles/.
set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || exit 1
timeout -k 10 300 bash tools/prof_bench.sh r3f || exit 1
timeout -k 10 300 bash tools/prof_workload.sh cifar10 || exit 1
timeout -k 10 300 bash tools/prof_workload.sh celeba64 || exit 1
timeout -k 10 200 python bench.py --workload mnist_c1 --no-cpu-baseline > gpurun_out/r03_bench_c1.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --force-dp --no-cpu-baseline > gpurun_out/r03_bench_forcedp.json 2> gpurun_out/r03_bench_forcedp.err || exit 1
timeout -k 10 200 python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r03_bench_f16.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r03_bench_f32.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload celeba64 --dtype f16 --no-cpu-baseline > gpurun_out/r03_bench_c5_f16.json 2>/dev/null || exit 1
echo ALLDONE
