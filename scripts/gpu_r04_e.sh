#!/bin/bash
# round 4, GPU call e: VIF per-point kernels (one wavefront per point, chunked staging), u renewed after a new response, GPBoost iterations through route B
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_e; mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_vif.py tests/test_vecchia_gpu.py -m gpu -x -q) > $O/pytest_vif_vecchia.log 2>&1; grep -v "^$" $O/pytest_vif_vecchia.log | tail -12 | cut -c1-250
timeout 300 python scripts/gpu_vif_bench.py > $O/vif_bench.json 2> $O/vif_bench.err; cat $O/vif_bench.json; tail -3 $O/vif_bench.err
(time timeout 900 python scripts/gpu_routeB.py --gpboost-only --skip-cpu-1e6) > $O/routeB_gpboost.log 2>&1; grep -v "^$" $O/routeB_gpboost.log | grep -v "Info\] \(Total\|Number\|Start\)" | tail -12 | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_vif -- python $GRAFT_REPO_ROOT/scripts/gpu_vif_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_vif > $O/prof_vif_summary.txt 2>&1; head -16 $O/prof_vif_summary.txt | cut -c1-230; rm -rf $O/prof_vif
ls -la $O
