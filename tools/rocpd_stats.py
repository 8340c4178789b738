#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish).

    python tools/rocpd_stats.py gpurun_out/prof1/rec_results.db > profiles/r01_rec_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("sa::", "").replace("unsigned short", "bf16")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else ("display_name" if "display_name" in sym_cols else sym_cols[-1])
    q = f"""select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col} order by 3 desc"""
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, mn, mx in rows:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
