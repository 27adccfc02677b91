#!/bin/bash
# round 4, GPU call i: the children's search sums the chunk partials itself (no reduce launch per split): tree tests + timing
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_i; mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_hist_gpu.py tests/test_multirank_gpu.py -m gpu -x -q) > $O/pytest_hist_multirank.log 2>&1; grep -v "^$" $O/pytest_hist_multirank.log | grep -v "version\|Hostname\|Librccl" | tail -15 | cut -c1-300
timeout 300 python scripts/gpu_boost_iter.py > $O/boost_iter.log 2>&1; tail -25 $O/boost_iter.log | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_boost -- python $GRAFT_REPO_ROOT/scripts/gpu_boost_iter.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_boost > $O/prof_boost_iter_summary.txt 2>&1; head -14 $O/prof_boost_iter_summary.txt | cut -c1-230; rm -rf $O/prof_boost
(time timeout 900 python -m pytest tests/test_routes_gpu.py -m gpu -x -q) > $O/pytest_routes.log 2>&1; grep -v "^$" $O/pytest_routes.log | grep -v "version\|Hostname\|Librccl" | tail -8 | cut -c1-300
