#!/usr/bin/env python
"""Per-kernel means of the PMC counter(s) in a rocprofv3 rocpd database (one `--pmc` pass = one database).

    python tools/rocpd_pmc.py gpurun_out/r01d_fetch.db [gpurun_out/r01d_write.db ...] > profiles/r01_d_hbm_traffic.md

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch. On gfx950 FETCH_SIZE counts 128-byte requests
as 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section): the `x2` column applies that correction.
"""
import re
import sqlite3
import sys


def table(c, prefix):
    return [r[0] for r in c.execute("select name from sqlite_master where type='table'") if r[0].startswith(prefix)][0]


def load(path):
    c = sqlite3.connect(path)
    pe, ip, kd, ks = (table(c, p) for p in ("rocpd_pmc_event", "rocpd_info_pmc", "rocpd_kernel_dispatch", "rocpd_info_kernel_symbol"))
    q = f"""select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value), avg(d.end - d.start)
            from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id
            join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name"""
    return list(c.execute(q))


def short(name):
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).replace("sa::", "")
    return name[:100]


def main(paths):
    rows = [r for p in paths for r in load(p)]
    rows.sort(key=lambda r: -r[4])
    print("| kernel | counter | dispatches | mean KiB / dispatch | x2 (gfx950 read correction) MB | total MB | mean us (under PMC) |")
    print("|---|---|---|---|---|---|---|")
    for name, ctr, n, mean, tot, dur in rows:
        if tot <= 0:
            continue
        x2 = f"{mean * 2 * 1024 / 1e6:.3f}" if ctr == "FETCH_SIZE" else "-"
        print(f"| `{short(name)}` | {ctr} | {n} | {mean:.1f} | {x2} | {tot * 1024 / 1e6:.1f} | {dur / 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1:])
