export TMPDIR=/tmp
OUT=gpurun_out/prof_lap1e6; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o lap -- python scripts/gpu_laplace.py 1000000 30 > $OUT/run.log 2> $OUT/trace.err
tail -2 $OUT/run.log
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/prof_lap1e6/trace/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for r in c.execute("select substr(name,1,100), count(*), sum(duration)/1e6, avg(duration)/1e3 from kernels group by name order by 3 desc limit 14"): print("%-100s n=%7d sum_ms=%9.1f avg_us=%8.1f" % r)
PY
rm -rf gpurun_out/prof_lap1e6/trace
