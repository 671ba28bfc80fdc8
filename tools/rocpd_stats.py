#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (the .db rocprofv3 7.2 writes by default): name, calls, total us, avg us, %."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
q = f"""select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from rocpd_kernel_dispatch{sfx} d
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"""
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print("total kernel time %.1f ms (/%g = %.2f ms)" % (tot / 1e3, div, tot / 1e3 / div))
print("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("| %s | %d | %.1f | %.2f | %.1f |" % (r[0][:72], r[1], r[3], r[2] / 1e3, 100 * r[2] / tot))
