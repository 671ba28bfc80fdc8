#!/usr/bin/env python3
"""Per (kernel, grid size) totals from a rocprofv3 rocpd database: separates the call sites that share one kernel instantiation
(e.g. the data-gradient GEMMs of the WaveGlow backward).  usage: rocpd_by_grid.py results.db [divide-by] [rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
cols = [r[1] for r in db.execute(f"pragma table_info(rocpd_kernel_dispatch{sfx})")]
gx = "d.grid_size_x" if "grid_size_x" in cols else "d.grid_x"
wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else "d.workgroup_x"
q = f"""select s.kernel_name, {gx}/{wx}, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from rocpd_kernel_dispatch{sfx} d
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id group by s.kernel_name, {gx}/{wx} order by 4 desc"""
rows = list(db.execute(q))
tot = sum(r[3] for r in rows)
print("total kernel time %.1f ms (/%g = %.2f ms)" % (tot / 1e3, div, tot / 1e3 / div))
print("| kernel | workgroups | calls | avg us | total ms/step | % |\n|---|---|---|---|---|---|")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("| %s | %d | %d | %.1f | %.2f | %.1f |" % (r[0][:60], r[1], r[2], r[4], r[3] / 1e3 / div, 100 * r[3] / tot))
