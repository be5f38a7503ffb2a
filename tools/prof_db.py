#!/usr/bin/env python
"""Per-kernel summary from a rocprofv3 rocpd SQLite database (…_results.db): calls, ms/step, average us."""
import sqlite3, sys
db, steps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"
rows = list(c.execute(q)); tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / steps:.2f} ms/step over {steps} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r[0][:110]:110s} {r[1]:6d} {r[2] / steps:8.2f} ms/step {r[3]:9.1f} us")
