#!/bin/bash
# usage (on the GPU box): tools/pmc_dominant.sh <tag>
# Three SEPARATE counter passes (counters only: --pmc with --kernel-trace, no other trace domain) over the two dense
# 5x5 shapes of the bench workload -> gpurun_out/<tag>_pmc_{dominant,wave_states,wgrad_wave_states}.json
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_util -o r -- python3 $R/tools/bench_conv.py --only=12,13 > $R/gpurun_out/pmc_${tag}_util.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_fetch -o r -- python3 $R/tools/bench_conv.py --only=12,13 > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_write -o r -- python3 $R/tools/bench_conv.py --only=12,13 > $R/gpurun_out/pmc_${tag}_write.log 2>&1
cd $R
python tools/make_pmc_util_json.py gpurun_out/pmc_${tag}_util gpurun_out/${tag}_pmc_wave_states.json k_conv_halo > /dev/null
python tools/make_pmc_util_json.py gpurun_out/pmc_${tag}_util gpurun_out/${tag}_pmc_wgrad_wave_states.json k_wgrad_halo > /dev/null
python tools/make_pmc_json.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_dominant.json k_conv_halo > /dev/null
for d in util fetch write; do
  f=$(ls gpurun_out/pmc_${tag}_$d/*/*counter_collection.csv gpurun_out/pmc_${tag}_$d/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|halo" "$f" > gpurun_out/${tag}_pmc_${d}.csv
  [ -n "$f" ] && cut -d, -f9 "$f" | cut -c1-60 | sort | uniq -c > gpurun_out/${tag}_pmc_${d}_kernels.txt
  rm -rf gpurun_out/pmc_${tag}_$d
done
grep -h "mfma_util\|frac_\|ratio" gpurun_out/${tag}_pmc_wave_states.json gpurun_out/${tag}_pmc_wgrad_wave_states.json gpurun_out/${tag}_pmc_dominant.json
