#!/bin/bash
# usage (on the GPU box): tools/prof_workload.sh <workload> [steps] [warmup] -> gpurun_out/prof_<workload>_summary.txt
w=$1; steps=${2:-5}; warm=${3:-2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o r -- python3 $R/bench.py --workload $w --steps $steps --warmup $warm > $R/gpurun_out/bench_$w.log 2>&1
grep -h "^{" $R/gpurun_out/bench_$w.log | cut -c1-260
cd $R && python tools/prof_summary.py gpurun_out/prof_$w $((steps + warm + 1)) --grid > gpurun_out/prof_${w}_summary.txt
rm -f gpurun_out/prof_$w/*kernel_trace.csv gpurun_out/prof_$w/*.db
