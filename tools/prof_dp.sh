#!/bin/bash
# usage (on the GPU box): tools/prof_dp.sh <tag> [steps]  -> gpurun_out/prof_<tag>_summary.txt : kernel trace of bench.py --force-dp
tag=$1; steps=${2:-10}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o r -- python3 $R/bench.py --steps $steps --warmup 3 --no-cpu-baseline --force-dp > $R/gpurun_out/bench_$tag.log 2>&1
grep -h "^{" $R/gpurun_out/bench_$tag.log | cut -c1-200
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag $((2 * (steps + 3) + 6)) > gpurun_out/prof_${tag}_summary.txt
rm -f gpurun_out/prof_$tag/*kernel_trace.csv gpurun_out/prof_$tag/*.db
