#!/usr/bin/env python3
"""Per-kernel PMC counter summary from a rocprofv3 rocpd database: mean counter value per dispatch of each kernel."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
pe_cols = [r[1] for r in db.execute(f"pragma table_info({pe})")]
pi_cols = [r[1] for r in db.execute(f"pragma table_info({pi})")]
kd_cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
ev_key = "event_id" if "event_id" in pe_cols else pe_cols[1]
kd_ev = "event_id" if "event_id" in kd_cols else "id"
q = f"""select s.kernel_name, i.name, d.id, sum(e.value), (d.end - d.start)
        from {pe} e join {pi} i on e.pmc_id = i.id join {kd} d on e.{ev_key} = d.{kd_ev}
        join {ks} s on d.kernel_id = s.id group by d.id, i.name"""
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for kname, cname, did, val, dt in db.execute(q):
    acc[kname][cname].append(val)
    dur[kname].append(dt)
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    print(k[:100])
    print(f"    dispatches {len(set(dur[k]))}  mean duration {sum(dur[k])/len(dur[k])/1e3:.1f} us")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:34s} mean/dispatch {sum(v)/len(v):16.1f}")
