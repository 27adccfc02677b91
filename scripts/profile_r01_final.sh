#!/bin/bash
# Final round-1 evidence run (on the GPU box via gpurun): full GPU test suite, default bench, rocprofv3 kernel trace of the bench and of
# the config-4 path, PCIe-inclusive rate.  Outputs under gpurun_out/final/.
export TMPDIR=/tmp
OUT=gpurun_out/final; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 400 $OUT/bench_default.json; echo
timeout 120 python scripts/gpu_pcie.py > $OUT/pcie.log 2>&1; tail -1 $OUT/pcie.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_bench -o bench -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace_bench.err
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_lap -o lap -- python scripts/gpu_laplace.py 100000 30 > $OUT/laplace_traced.log 2> $OUT/trace_lap.err
timeout 100 python scripts/gpu_laplace.py 100000 30 > $OUT/laplace.log 2>&1; tail -2 $OUT/laplace.log
timeout 100 python scripts/fit_bench.py --n 100000 > $OUT/fit_n1e5.json 2>&1; tail -n 1 $OUT/fit_n1e5.json | head -c 300; echo
timeout 100 python scripts/fit_bench.py --n 1000000 > $OUT/fit_n1e6.json 2>&1; tail -n 1 $OUT/fit_n1e6.json | head -c 300; echo
timeout 100 python scripts/gpu_boost_iter.py 100000 8 > $OUT/boost_iter_n1e5.json 2>&1; tail -n 1 $OUT/boost_iter_n1e5.json | tail -c 420; echo
timeout 100 python scripts/gpu_boost_iter.py 1000000 8 > $OUT/boost_iter_n1e6.json 2>&1; tail -n 1 $OUT/boost_iter_n1e6.json | tail -c 420; echo
timeout 100 python scripts/gpu_hist_bench.py > $OUT/hist_bench.log 2>&1; cat $OUT/hist_bench.log
timeout 100 python scripts/gpu_exact_bench.py 2000 16384 > $OUT/exact_bench.log 2>&1; cat $OUT/exact_bench.log
python - <<'PY'
import sqlite3, glob
for tag in ("trace_bench", "trace_lap"):
    dbs = glob.glob('gpurun_out/final/%s/**/*.db' % tag, recursive=True)
    if not dbs: print(tag, "no db"); continue
    c = sqlite3.connect(dbs[0])
    with open('gpurun_out/final/%s_summary.txt' % tag, 'w') as f:
        f.write("== rocprofv3 --kernel-trace --stats (%s): per-kernel calls, total ms, mean us, min us, max us; VGPR, LDS ==\n" % tag)
        for r in c.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc"):
            f.write("%-110s calls=%7d total_ms=%10.2f mean_us=%10.2f min_us=%9.2f max_us=%10.2f vgpr=%s lds=%s\n" % ((str(r[0])[:110],) + tuple(r[1:])))
    print(open('gpurun_out/final/%s_summary.txt' % tag).read()[:1500])
PY
rm -rf $OUT/trace_bench $OUT/trace_lap
