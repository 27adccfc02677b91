#!/bin/bash
# rocprofv3 kernel trace of the config-3 boosting iteration (scripts/gpu_boost_iter.py): which kernels the tree's 0.36 ms per split go to.
export TMPDIR=/tmp
OUT=gpurun_out/tree; mkdir -p $OUT
N=${1:-100000}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python scripts/gpu_boost_iter.py $N > $OUT/run.log 2> $OUT/trace.err
python scripts/summarize_prof.py trace $OUT/trace > $OUT/summary.txt; head -24 $OUT/summary.txt; tail -1 $OUT/run.log
rm -rf $OUT/trace
