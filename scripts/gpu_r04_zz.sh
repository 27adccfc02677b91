#!/bin/bash
# round 4, the LAST GPU call: after the route-B Laplace seams (patch, tests and scripts only -- gpboost_amd/, include/ and bench.py are unchanged since
# scripts/gpu_r04_z.sh ran) the whole -m gpu suite, smoke() and the default bench line once more at HEAD
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_zz; mkdir -p $O
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; grep -v "^$" $O/pytest_gpu.log | grep -v "version\|Hostname\|Librccl" | tail -25 | cut -c1-300
(time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')") > $O/smoke.log 2>&1; tail -5 $O/smoke.log | cut -c1-300
(time timeout 300 python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; head -c 700 $O/bench_default.json; tail -3 $O/bench_default.err | cut -c1-200
