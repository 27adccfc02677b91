#!/bin/bash
# round 4, GPU call p: route B after the last seam change (device gradient re-finds the mode after a reset): Laplace seams + GPBoost for binary data
# against the stored CPU values, and the three route tests
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_p; mkdir -p $O
export TMPDIR=/tmp
(time timeout 400 python scripts/gpu_routeB.py --laplace-only) > $O/routeB_laplace.log 2>&1; grep -v "^$" $O/routeB_laplace.log | grep -v "Info\] \(Total\|Number\|Start\)" | tail -14 | cut -c1-330
(time timeout 600 python -m pytest tests/test_routes_gpu.py -m gpu -q -p no:cacheprovider) > $O/pytest_routes.log 2>&1; grep -v "^$" $O/pytest_routes.log | grep -v "version\|Hostname\|Librccl" | tail -8 | cut -c1-300
