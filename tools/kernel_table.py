#!/usr/bin/env python3
"""Per-kernel / per-grid table from a rocprofv3 rocpd database (kernel trace):
usage: tools/kernel_table.py <results.db> [name-substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pats = sys.argv[2:] or ['']
where = ' or '.join("name like '%%%s%%'" % p for p in pats)
q = ("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), min(duration) "
     "from kernels where %s group by name, grid_x, grid_y, grid_z order by name" % where)
for r in cur.execute(q):
    print(f"{r[0][:70]:70} grid {r[1] // r[4]:>6}x{r[2]}x{r[3]:<3} wg {r[4]:4d} n={r[5]:4d} "
          f"avg {r[6] / 1e3:7.1f} us min {r[7] / 1e3:7.1f}")
