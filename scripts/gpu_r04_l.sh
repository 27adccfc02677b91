#!/bin/bash
# round 4, GPU call l / m: route B Laplace seams (Likelihood::FindModePostRandEffCalcMLLVecchia / gradient / ResetModeToPreviousValue on the device)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_m; mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python scripts/gpu_routeB.py --laplace-only) > $O/routeB_laplace.log 2>&1; grep -v "^$" $O/routeB_laplace.log | grep -v "Info\] \(Total\|Number\|Start\)" | tail -25 | cut -c1-330
