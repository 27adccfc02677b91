#!/bin/bash
# PMC passes (separate runs, no trace domains) for the histogram build at n = 1e7: HBM traffic of hist_build_kernel.
export TMPDIR=/tmp
OUT=gpurun_out/histpmc; mkdir -p $OUT
B="python scripts/gpu_hist_bench.py"
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o h -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o h -- $B > /dev/null 2> $OUT/pmc_write.err
python scripts/summarize_prof.py $OUT | grep -i "hist" > $OUT/summary.txt; cat $OUT/summary.txt | head -40
rm -rf $OUT/pmc_fetch $OUT/pmc_write
