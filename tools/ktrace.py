"""Per-kernel summary (count, mean, min in us) of a rocprofv3 --kernel-trace database (rocpd .db)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 2*3 desc"
for r in c.execute(q):
    if r[1] >= int(sys.argv[2]) if len(sys.argv) > 2 else True:
        print(r[0][:80].ljust(80), r[1], round(r[2], 2), round(r[3], 2))
