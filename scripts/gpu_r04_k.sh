#!/bin/bash
# round 4, GPU call k: rehearsal of the final evidence run -- whole -m gpu suite, smoke(), default bench line
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k; mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu_full.log 2>&1; grep -v "^$" $O/pytest_gpu_full.log | grep -v "version\|Hostname\|Librccl" | tail -30 | cut -c1-300
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')") > $O/smoke.log 2>&1; tail -3 $O/smoke.log | cut -c1-300
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err | cut -c1-300
