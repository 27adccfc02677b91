#!/bin/bash
# round 4, GPU call b: device k-means (VIF set-up), mailbox (multi-rank threads + forced-dist bench), Laplace block solve v2 (16-byte sc1 gathers)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vif.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_vif.log; cat $O/pytest_vif.log
timeout 300 python scripts/gpu_vif_bench.py > $O/vif_bench.log 2>&1; tail -2 $O/vif_bench.log
timeout 600 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q -k "mailbox or three_ranks" 2>&1 | tail -15 > $O/pytest_mailbox.log; cat $O/pytest_mailbox.log
for w in 64 128 192 256; do
  GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_WGS=$w timeout 300 python scripts/gpu_laplace.py > $O/laplace_sf5v2_w$w.log 2>&1; tail -2 $O/laplace_sf5v2_w$w.log
done
GPB_LAP_SYNCFREE=5 timeout 600 python -m pytest tests/test_laplace_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_laplace_sf5v2.log; cat $O/pytest_laplace_sf5v2.log
# one evaluation's cost outside the kernel: single GPU; 1-rank RCCL in the loop; 1-rank mailbox in the loop
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_single.json 2> $O/bench_single.err; tail -c 600 $O/bench_single.err
GPB_BENCH_FORCE_DIST=1 GPB_BENCH_NO_MAILBOX=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_dist_rccl.json 2> $O/bench_forced_dist_rccl.err; tail -c 600 $O/bench_forced_dist_rccl.err
GPB_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_forced_dist_mailbox.json 2> $O/bench_forced_dist_mailbox.err; tail -c 600 $O/bench_forced_dist_mailbox.err
python - <<'P'
import json
for f in ("bench_single", "bench_forced_dist_rccl", "bench_forced_dist_mailbox"):
    try:
        d = json.loads([l for l in open("gpurun_out/r04_b/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["overhead_us"], d["config"].get("rccl_ranks"), d["config"].get("mailbox_ranks"), d["config"].get("kernel_ms_source"))
    except Exception as e:
        print(f, "failed", e)
P
cd /tmp
GPB_LAP_SYNCFREE=5 GPB_LAP_SFW_WGS=128 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lap5 -- python $GRAFT_REPO_ROOT/scripts/gpu_laplace.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_prof.py trace $O/prof_lap5 > $O/prof_lap5v2_summary.txt 2>&1; head -14 $O/prof_lap5v2_summary.txt | cut -c1-230; rm -rf $O/prof_lap5
ls -la $O
