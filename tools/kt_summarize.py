#!/usr/bin/env python3
"""Per-kernel statistics of a `rocprofv3 --kernel-trace --stats` results database (the same numbers as its *_kernel_stats.csv):
    python tools/kt_summarize.py <dir with *_results.db> [title] [last N]  > profiles/<name>.txt
With `last N` a second table covers only the last N dispatches of every kernel (bench.py: the N steps of the timed region — the earlier
launches are the host-entry measurement and the warm-up, run while the device clocks are still settling)."""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True))
assert dbs, "no results database under " + src
print(f"# {title}")
print(f"# source: rocprofv3 --kernel-trace --stats, {os.path.relpath(dbs[0])}")
c = sqlite3.connect(dbs[0])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
for n, cnt, s, a, mn, mx in rows:
    print(f"{cnt:>6} {s / 1e3:>12.1f} {a / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * s / tot:>6.1f}  {n}")

last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if last > 0:
    print()
    print(f"# the last {last} dispatches of every kernel (the timed region of bench.py)")
    print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for n, cnt, *_ in rows:
        d = [r[0] for r in c.execute("select duration from kernels where name = ? order by start", (n,))][-last:]
        if len(d) >= last:
            print(f"{len(d):>6} {sum(d) / len(d) / 1e3:>10.2f} {min(d) / 1e3:>10.2f} {max(d) / 1e3:>10.2f}  {n}")
