#!/bin/bash
# round 4, GPU call c: Laplace block solve v3 (J = chunks per 16-lane group) sweep; histogram rows kernel without the padding features
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_c; mkdir -p $O
export TMPDIR=/tmp
for j in 1 2 4; do for w in 256 512 768 1024; do
  GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_J=$j GPB_LAP_SFW_WGS=$w timeout 300 python scripts/gpu_laplace.py > $O/laplace_sfw3_j${j}_w$w.log 2>&1; echo "J=$j W=$w"; tail -1 $O/laplace_sfw3_j${j}_w$w.log | cut -c1-330
done; done
GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_J=1 timeout 600 python -m pytest tests/test_laplace_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_laplace_j1.log; cat $O/pytest_laplace_j1.log
GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_J=2 timeout 600 python -m pytest tests/test_laplace_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_laplace_j2.log; cat $O/pytest_laplace_j2.log
timeout 600 python -m pytest tests/test_hist_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_hist.log; cat $O/pytest_hist.log
timeout 300 python scripts/gpu_hist_bench.py > $O/hist_bench.log 2>&1; tail -12 $O/hist_bench.log
ls -la $O
