"""Condenses rocprofv3 output (rocpd sqlite: kernel trace + PMC passes) into a small text summary for profiles/."""
import glob, os, sqlite3, sys
out = sys.argv[1]


def q(db, sql):
    try:
        return sqlite3.connect(db).execute(sql).fetchall()
    except Exception as e:
        return [("ERR", str(e))]


tr = glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)
print("== kernel stats (rocprofv3 --kernel-trace --stats; view top_kernels) ==")
for db in tr:
    for r in q(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-100s calls=%s total_ns=%s avg_ns=%.1f pct=%.2f" % (str(r[0])[:100], r[1], r[2], r[3], r[4]))
    print("== per-kernel resources ==")
    for r in q(db, "select name,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,grid_x,workgroup_x,count(*),avg(duration),min(duration),max(duration) from kernels group by name"):
        print("%-100s VGPR=%s AGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s n=%s avg_ns=%.1f min=%s max=%s" % ((str(r[0])[:100],) + tuple(r[1:])))
print("== PMC: mean per dispatch, by kernel ==")
for db in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True)):
    for r in q(db, "select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
        if "vecchia_point" in str(r[0]) or "hist" in str(r[0]) or "nn_kernel" in str(r[0]):
            print("%-60s %-24s mean=%.6g n=%s" % (str(r[0])[:60], r[1], r[2], r[3]))
