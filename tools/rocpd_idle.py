#!/usr/bin/env python3
"""GPU idle time from a rocprofv3 rocpd database: over the last fraction F of the trace (default 0.5: the timed steps), the union
of all kernels' [start, end) intervals against the span, and the largest gaps with the kernel that ends / starts each - where
the device waits for the host.  Usage: rocpd_idle.py db [F=0.5] [N=25]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = list(db.execute(f"""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{sfx} d
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id order by d.start"""))
t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
cut = t_hi - frac * (t_hi - t_lo)
rows = [r for r in rows if r[1] >= cut]
busy, gaps = 0, []
cur_s, cur_e, last_name = rows[0][1], rows[0][2], rows[0][0]
for name, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - rows[0][1], last_name, name))
        cur_s, cur_e, last_name = s, e, name
    elif e > cur_e:
        cur_e, last_name = e, name
busy += cur_e - cur_s
span = cur_e - rows[0][1]
print("span %.2f ms, busy (union over queues) %.2f ms, idle %.2f ms (%.1f %%), %d gaps" %
      (span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, len(gaps)))
small = sum(g[0] for g in gaps if g[0] < 3000)
print("gaps < 3 us: %.2f ms in %d gaps; 3-20 us: %.2f ms; > 20 us: %.2f ms" %
      (small / 1e6, sum(1 for g in gaps if g[0] < 3000), sum(g[0] for g in gaps if 3000 <= g[0] < 20000) / 1e6,
       sum(g[0] for g in gaps if g[0] >= 20000) / 1e6))
print("| gap us | at ms | after kernel | before kernel |\n|---|---|---|---|")
for g in sorted(gaps, reverse=True)[:n]:
    short = lambda s: s.replace("_kernel", "").split("(")[0][:40]
    print("| %.1f | %.2f | %s | %s |" % (g[0] / 1e3, g[1] / 1e6, short(g[2]), short(g[3])))
