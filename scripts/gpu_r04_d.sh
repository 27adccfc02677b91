#!/bin/bash
# round 4, GPU call d: block solve v3 as the default (all Laplace test files), histogram / tree tests, route B with the widened seams, MFMA ubench
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_d; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/gpu_laplace.py > $O/laplace_default.log 2>&1; tail -3 $O/laplace_default.log | cut -c1-330
(time timeout 1200 python -m pytest tests/test_laplace_gpu.py tests/test_z_laplace_grad_gpu.py tests/test_laplace_dup.py tests/test_laplace_predvar.py tests/test_zz_laplace_train_re_gpu.py -m gpu -x -q) > $O/pytest_laplace_all.log 2>&1; grep -v "^$" $O/pytest_laplace_all.log | tail -8 | cut -c1-250
(time timeout 900 python -m pytest tests/test_hist_gpu.py tests/test_multirank_gpu.py -m gpu -x -q) > $O/pytest_hist_multirank.log 2>&1; grep -v "^$" $O/pytest_hist_multirank.log | grep -v "version\|Hostname\|Librccl" | tail -8 | cut -c1-250
cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -I../../gpboost_amd/csrc ldlt_mfma44.hip -o ldlt_mfma44 2>/dev/null; cd ../..
timeout 120 scripts/ubench/ldlt_mfma44 > $O/ubench_ldlt_mfma44.log 2>&1; cat $O/ubench_ldlt_mfma44.log
(time timeout 1500 python scripts/gpu_routeB.py) > $O/routeB.log 2>&1; grep -v "^$" $O/routeB.log | tail -40 | cut -c1-330
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lap -- python $GRAFT_REPO_ROOT/scripts/gpu_laplace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_lap > $O/prof_laplace_default_summary.txt 2>&1; head -16 $O/prof_laplace_default_summary.txt | cut -c1-230; rm -rf $O/prof_lap
ls -la $O
