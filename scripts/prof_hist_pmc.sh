#!/bin/bash
# Histogram build at n = 1e7 (scripts/gpu_hist_bench.py): kernel trace (durations of hist_build_kernel / hist_reduce_kernel) and PMC
# passes, each in its own run without trace domains: HBM traffic, LDS conflict / wait counters, VALU activity.
# Usage: scripts/prof_hist_pmc.sh <tag>  -> gpurun_out/<tag>/{trace_summary.txt,pmc_summary.txt}
export TMPDIR=/tmp
TAG=${1:-histpmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python scripts/gpu_hist_bench.py"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_trace.log 2> $OUT/trace.err
python scripts/summarize_prof.py trace $OUT/trace > $OUT/trace_summary.txt; head -8 $OUT/trace_summary.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -- $B > $OUT/pmc$i.log 2> $OUT/pmc$i.err || echo "pmc pass $i ($C) failed: $(tail -2 $OUT/pmc$i.err)"
done
python scripts/summarize_prof.py pmc $OUT/pmc* | grep -i "hist_" > $OUT/pmc_summary.txt; cat $OUT/pmc_summary.txt | awk '{print $(NF-2), $(NF-1), $NF}' | head -60
rm -rf $OUT/pmc[0-9] $OUT/trace
