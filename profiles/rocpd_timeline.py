#!/usr/bin/env python3
"""Timeline (start offset, duration, stream/queue) of the kernel dispatches of the LAST n steps in a rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else cols[0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
rows = rows[-n:]
t0 = rows[0][0]
for st, en, name, q in rows:
    print(f"{(st-t0)/1e3:9.1f} us  +{(en-st)/1e3:8.1f} us  q{q}  {name[:70]}")
