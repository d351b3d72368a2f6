#!/usr/bin/env python3
"""Per-kernel summary (count / total / avg / min / max, microseconds) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats ...` writes *_results.db on this image)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
q = f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for name, n, s, a, mn, mx in rows:
    print(f"{name[:90]:90s} {n:6d} {s/1e3:10.1f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}")
