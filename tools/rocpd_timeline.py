#!/usr/bin/env python3
"""A window of the kernel timeline from a rocprofv3 rocpd database: for N consecutive dispatches starting at the first launch of
kernel <pattern> after fraction F of the trace: queue, start offset (us), duration (us), gap to the previous kernel's end on the
same queue (us).  Usage: rocpd_timeline.py db pattern [F=0.5] [N=40]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
cols = [r[1] for r in db.execute(f"pragma table_info(rocpd_kernel_dispatch{sfx})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute(f"""select s.kernel_name, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch{sfx} d
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id order by d.start"""))
t_lo, t_hi = rows[0][1], rows[-1][2]
cut = t_lo + frac * (t_hi - t_lo)
i0 = next(i for i, r in enumerate(rows) if r[1] >= cut and pat in r[0])
last_end = {}
for r in rows[max(0, i0 - 30):i0]:
    last_end[r[3]] = r[2]
base = rows[i0][1]
print("| queue | kernel | start us | dur us | gap on queue us |\n|---|---|---|---|---|")
for r in rows[i0:i0 + n]:
    gap = (r[1] - last_end[r[3]]) / 1e3 if r[3] in last_end else float("nan")
    name = r[0].replace("_kernel", "").split("(")[0][:44]
    print("| %s | %s | %.1f | %.1f | %.1f |" % (r[3], name, (r[1] - base) / 1e3, (r[2] - r[1]) / 1e3, gap))
    last_end[r[3]] = r[2]
