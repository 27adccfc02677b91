#!/bin/bash
# round 4, LAST GPU call: the evidence set at HEAD -- whole -m gpu suite, smoke(), PMC passes + kernel trace (scripts/profile_r04.sh), default bench line,
# config 5's shape, the mailbox path with one rank in the loop
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_z; mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; grep -v "^$" $O/pytest_gpu.log | grep -v "version\|Hostname\|Librccl" | tail -25 | cut -c1-300
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')") > $O/smoke.log 2>&1; tail -6 $O/smoke.log | cut -c1-400
bash scripts/profile_r04.sh r04_z > $O/profile_r04.log 2>&1; tail -30 $O/profile_r04.log | cut -c1-260
# the PMC means of THIS run are what the bench lines below read roofline.traffic from (the same file is committed as profiles/r04_pmc.json)
cp $O/pmc.json profiles/r04_pmc.json
(time timeout 900 python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 --d 3 --cov matern_2.5 --m 40 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err; head -c 1200 $O/bench_config5.json; tail -3 $O/bench_config5.err | cut -c1-300
GPB_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_forced_dist_mailbox.json 2> $O/bench_forced_dist_mailbox.err; head -c 1500 $O/bench_forced_dist_mailbox.json; tail -3 $O/bench_forced_dist_mailbox.err | cut -c1-300
ls -la $O
