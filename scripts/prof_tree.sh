#!/bin/bash
# rocprofv3 kernel trace of the config-3 boosting iteration (scripts/gpu_boost_iter.py): which kernels the tree's 0.36 ms per split go to.
export TMPDIR=/tmp
OUT=gpurun_out/tree; mkdir -p $OUT
N=${1:-100000}
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o tree -- python scripts/gpu_boost_iter.py $N > $OUT/run.log 2> $OUT/trace.err
python - <<'PY'
import sqlite3, glob
dbs = glob.glob('gpurun_out/tree/trace/**/*.db', recursive=True)
c = sqlite3.connect(dbs[0])
with open('gpurun_out/tree/summary.txt', 'w') as f:
    f.write("== rocprofv3 --kernel-trace --stats: scripts/gpu_boost_iter.py (3 iterations + 1 harness tree): calls, total ms, mean us ==\n")
    for r in c.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels group by name order by 3 desc"):
        f.write("%-100s calls=%6d total_ms=%9.3f mean_us=%9.2f min_us=%8.2f max_us=%9.2f\n" % ((str(r[0])[:100],) + tuple(r[1:])))
print(open('gpurun_out/tree/summary.txt').read()[:4000])
PY
rm -rf $OUT/trace
