#!/usr/bin/env python3
"""HBM-side bytes per launch of every kernel from two rocprofv3 PMC passes (separate runs, as the MI355X guide prescribes):

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tacotron
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tacotron
  python3 tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db > profiles/<name>.json

Counter unit is KiB.  FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md); the
calibration kernel is pack_table_kernel, which by construction reads 1073 MB of f32 parameters and writes 1073 MB of planes.
"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    sfx = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_pmc_event%'")][0].replace("rocpd_pmc_event", "")
    q = f"""select s.kernel_name, count(*), sum(e.value) from rocpd_pmc_event{sfx} e
            join rocpd_info_pmc{sfx} p on e.pmc_id = p.id
            join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol{sfx} s on d.kernel_id = s.id
            where p.name = '{counter}' group by s.kernel_name"""
    return {name: (n, tot) for name, n, tot in db.execute(q)}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"how": __doc__.strip().split("\n\n")[0] + " | unit KiB, FETCH_SIZE doubled (gfx950 note); see the tool's docstring",
           "kernels": {}}
    for name in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, f = fetch[name]
        nw, w = write.get(name, (0, 0.0))
        if not n or name.startswith("_ZN2at") or "rocclr" in name:
            continue
        fb = 2.0 * f * 1024 / n
        wb = (w * 1024 / nw) if nw else 0.0
        out["kernels"][name] = {"launches": n, "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
