#!/usr/bin/env python3
"""Per-kernel averages of the counters in one or more rocprofv3 --pmc databases (rocpd SQLite).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d gpurun_out/pmc_a -o a -- python3 bench.py ...
    python3 tools/pmc_counters.py [--match SUBSTR] gpurun_out/pmc_a/a_results.db [more.db ...]

Prints JSON: {kernel: {"launches": n, counter: mean value per launch, ...}}.  Counters of several passes (separate runs, as the
MI355X guide prescribes: no trace domains next to --pmc) are merged per kernel name.
"""
import json
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_pmc_event%'")][0].replace("rocpd_pmc_event", "")
    q = f"""select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event{sfx} e
            join rocpd_info_pmc{sfx} p on e.pmc_id = p.id
            join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id
            group by s.kernel_name, p.name"""
    out = {}
    for kname, cname, n, tot in db.execute(q):
        k = out.setdefault(kname, {"launches": n})
        k[cname] = tot / n
    return out


def main():
    args = sys.argv[1:]
    match = None
    if args and args[0] == "--match":
        match = args[1]
        args = args[2:]
    merged = {}
    for path in args:
        for k, v in per_kernel(path).items():
            if match and match not in k:
                continue
            merged.setdefault(k, {}).update(v)
    json.dump(merged, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
