#!/usr/bin/env python3
"""Coarse phases of a Tacotron-2 training step from a rocprofv3 rocpd database: steps are cut at the Adam kernel; inside a step the
forward decoder loop is [first sbgemm_lstm, last att_softmax_ctx], the backward loop [first att_bwd_fused, last lstm_cell_bwd_q + its
GEMM].  Prints, per step, the length of each phase (ms) and the device-busy share (union over queues) inside it.
Usage: rocpd_phases.py db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = list(db.execute(f"""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{sfx} d
        join rocpd_info_kernel_symbol{sfx} s on d.kernel_id=s.id order by d.start"""))
cuts = [i for i, r in enumerate(rows) if "adam_table" in r[0]]


def busy(seg, lo, hi):
    ev = sorted((max(r[1], lo), min(r[2], hi)) for r in seg if r[2] > lo and r[1] < hi)
    tot, cur_s, cur_e = 0, None, None
    for s, e in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


print("| step | before fwd loop | fwd loop | between loops | bwd loop | after bwd loop (incl. Adam) | total |")
print("|---|---|---|---|---|---|---|")
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][2], rows[b][2]
    f0 = next(r[1] for r in seg if "sbgemm_lstm" in r[0])
    b0 = next(r[1] for r in seg if "att_bwd_fused" in r[0])
    # (the attention's last launch of the forward loop: softmax + context, or the energies launch that contains them)
    f1 = max(r[2] for r in seg if ("att_softmax_ctx" in r[0] or "att_energy" in r[0]) and r[1] < b0)
    last_att = max(x[2] for x in seg if "att_bwd_fused" in x[0])
    b1 = max(r[2] for r in seg if ("lstm_cell_bwd_q" in r[0] or "sbgemm_plain" in r[0]) and r[1] < last_att + 50000)
    ph = [(t0, f0), (f0, f1), (f1, b0), (b0, b1), (b1, t1)]
    cells = ["%.2f (%.0f %% busy)" % ((hi - lo) / 1e6, 100.0 * busy(seg, lo, hi) / max(1, hi - lo)) for lo, hi in ph]
    print("| %d | %s | %.2f |" % (cuts.index(a), " | ".join(cells), (t1 - t0) / 1e6))
