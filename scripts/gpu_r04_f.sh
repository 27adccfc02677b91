#!/bin/bash
# round 4, GPU call f: VIF per-point kernels with batched staging + MFMA Gram tiles (new cases m = 40 / 55 / 70), histogram flush variant A/B
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_f; mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_vif.py tests/test_laplace_gpu.py -m gpu -x -q) > $O/pytest_vif_laplace.log 2>&1; grep -v "^$" $O/pytest_vif_laplace.log | grep -v "version\|Hostname\|Librccl" | tail -12 | cut -c1-250
timeout 300 python scripts/gpu_vif_bench.py > $O/vif_bench.json 2> $O/vif_bench.err; cat $O/vif_bench.json; tail -3 $O/vif_bench.err
for v in default flushrw default flushrw; do
  if [ $v = flushrw ]; then export GPBOOST_AMD_LIB=$GRAFT_REPO_ROOT/gpboost_amd/csrc/build_alt/lib_alt_flushrw.so; else unset GPBOOST_AMD_LIB; fi
  echo "== hist bench: $v"; timeout 300 python scripts/gpu_hist_bench.py 2>&1 | grep -v "version\|Hostname\|Librccl" | tee -a $O/hist_bench_$v.log
done
unset GPBOOST_AMD_LIB
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_vif -- python $GRAFT_REPO_ROOT/scripts/gpu_vif_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_vif > $O/prof_vif_summary.txt 2>&1; head -12 $O/prof_vif_summary.txt | cut -c1-230; rm -rf $O/prof_vif
ls -la $O
