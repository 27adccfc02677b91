#!/bin/bash
# round 4, GPU call j: m = 40 gradient instance with one workgroup per CU (AGPRs, no scratch) against two (256 VGPRs + scratch); hist tests with the branch-free FixHistogram walk
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_j; mkdir -p $O
export TMPDIR=/tmp
for v in default onewg default onewg; do
  if [ $v = onewg ]; then export GPBOOST_AMD_LIB=$GRAFT_REPO_ROOT/gpboost_amd/csrc/build_alt/lib_alt_mt40_onewg.so; else unset GPBOOST_AMD_LIB; fi
  timeout 300 python scripts/gpu_grad_m40.py 2>&1 | grep -v "version\|Hostname\|Librccl" | tee -a $O/grad_m40_$v.log | cut -c1-400
done
unset GPBOOST_AMD_LIB
(time timeout 900 python -m pytest tests/test_hist_gpu.py -m gpu -x -q) > $O/pytest_hist.log 2>&1; grep -v "^$" $O/pytest_hist.log | grep -v "version\|Hostname\|Librccl" | tail -6 | cut -c1-300
timeout 300 python scripts/gpu_boost_iter.py > $O/boost_iter.log 2>&1; tail -1 $O/boost_iter.log | cut -c1-1200
