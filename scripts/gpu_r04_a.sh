#!/bin/bash
# round 4, GPU call a: first device run of the VIF gradient path (new row-major layout), Laplace probe-block solve A/B, kernel traces
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vif.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_a/pytest_vif.log
cat gpurun_out/r04_a/pytest_vif.log
timeout 300 python scripts/gpu_vif_bench.py > gpurun_out/r04_a/vif_bench.log 2>&1; tail -3 gpurun_out/r04_a/vif_bench.log
# Laplace: level launches for the block (1) vs wave-per-row sync-free block (5), workgroup counts
for v in 1 5; do
  GPB_LAP_SYNCFREE=$v timeout 300 python scripts/gpu_laplace.py > gpurun_out/r04_a/laplace_sf$v.log 2>&1; tail -4 gpurun_out/r04_a/laplace_sf$v.log
done
for w in 256 512 768; do
  GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_WGS=$w timeout 300 python scripts/gpu_laplace.py > gpurun_out/r04_a/laplace_sf5_w$w.log 2>&1; tail -2 gpurun_out/r04_a/laplace_sf5_w$w.log
done
GPB_LAP_SYNCFREE=5 timeout 600 python -m pytest tests/test_laplace_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r04_a/pytest_laplace_sf5.log; cat gpurun_out/r04_a/pytest_laplace_sf5.log
# kernel traces: VIF evaluation + gradient; Laplace with the new block solve
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_a/prof_vif -- python $GRAFT_REPO_ROOT/scripts/gpu_vif_bench.py > /dev/null 2>&1
GPB_LAP_SYNCFREE=5 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_a/prof_lap5 -- python $GRAFT_REPO_ROOT/scripts/gpu_laplace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for p in prof_vif prof_lap5; do
  python scripts/summarize_prof.py trace gpurun_out/r04_a/$p > gpurun_out/r04_a/${p}_summary.txt 2>&1
  head -32 gpurun_out/r04_a/${p}_summary.txt | cut -c1-230
  rm -rf gpurun_out/r04_a/$p
done
ls -la gpurun_out/r04_a
