#!/usr/bin/env python3
"""Per (kernel, grid) rows of a rocprofv3 rocpd database: calls, mean / min / max µs over the last 60 % of the trace -- which LAYER of a
kernel that serves many shapes is the slow one.  usage: prof_by_grid.py <db> <kernel-name substring> [...]"""
import sqlite3, sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
g = [k for k in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if k in cols][:3]
t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
cut = t0 + 0.4 * (t1 - t0)
for pat in sys.argv[2:]:
    rows = c.execute(f"select name, {', '.join(g)}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                     f"where name like ? and start > ? group by name, {', '.join(g)} order by 8 desc", (f"%{pat}%", cut)).fetchall()
    for r in rows:
        print(f"{r[0][:60]:60s} grid {str(tuple(r[1:4])):22s} x{r[4]:5d}  mean {r[5]/1e3:8.1f}  min {r[6]/1e3:8.1f}  max {r[7]/1e3:8.1f} us   total {r[8]/1e6:8.2f} ms")
