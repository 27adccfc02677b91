#!/bin/bash
# round 4, GPU call g / h: VIF per-point kernels with the k x k system in registers (readlane Cholesky / substitutions); h: batched loads of the low-rank dot products, 16 staged rows in flight
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_h; mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_vif.py -m gpu -q) > $O/pytest_vif.log 2>&1; grep -v "^$" $O/pytest_vif.log | grep -v "version\|Hostname\|Librccl" | tail -25 | cut -c1-300
timeout 300 python scripts/gpu_vif_bench.py > $O/vif_bench.json 2> $O/vif_bench.err; cat $O/vif_bench.json; tail -3 $O/vif_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_vif -- python $GRAFT_REPO_ROOT/scripts/gpu_vif_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_vif > $O/prof_vif_summary.txt 2>&1; head -12 $O/prof_vif_summary.txt | cut -c1-230; rm -rf $O/prof_vif
