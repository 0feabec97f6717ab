#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag> [steps] [warmup]   -> gpurun_out/prof_<tag>_summary.txt
tag=$1; steps=${2:-20}; warm=${3:-5}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o r -- python3 $R/bench.py --steps $steps --warmup $warm --no-cpu-baseline > $R/gpurun_out/bench_$tag.log 2>&1
grep -h "^{" $R/gpurun_out/bench_$tag.log | cut -c1-260
cd $R && python tools/prof_summary.py gpurun_out/prof_$tag $((steps + warm + 2)) --grid --json=gpurun_out/prof_${tag}_families.json > gpurun_out/prof_${tag}_summary.txt
rm -f gpurun_out/prof_$tag/*kernel_trace.csv gpurun_out/prof_$tag/*.db
