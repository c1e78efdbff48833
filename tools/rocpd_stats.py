"""Per-kernel summary of a rocprofv3 rocpd database (the default output of rocprofv3 in ROCm 7)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, sym))
rows = list(c.execute(q))
tot = sum(r[3] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (r[0][:110], r[1], r[3], r[2], r[4], r[5], 100.0 * r[3] / tot))
