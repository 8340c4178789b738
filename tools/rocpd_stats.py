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
    if "--gaps" in sys.argv:
        gaps(c, name_col)
    if "--by-grid" in sys.argv:
        by_grid(c, name_col, cols)


def by_grid(c, name_col, cols):
    """Per (kernel, grid size): average duration and average idle gap in front of the dispatch (start - previous end on the
    device). Separates launches of one template that differ only in shape (qkv / o / down split-K GEMMs)."""
    gcol = next((x for x in ("grid_size_x", "grid_x", "grid_size") if x in cols), None)
    if not gcol:
        print("\n(no grid-size column in this database; columns:", cols, ")")
        return
    q = f"""select d.start, d.end, s.{name_col}, d.{gcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
            on d.kernel_id = s.id order by d.start"""
    rows = list(c.execute(q))
    agg, prev_end = {}, None
    for st, en, name, grid in rows:
        k = (short(name)[:90], grid)
        a = agg.setdefault(k, [0, 0, 0, 0])
        a[0] += 1; a[1] += en - st
        if prev_end is not None and st - prev_end < 50_000:          # gaps >= 50 us are host stalls, not launch boundaries
            a[2] += st - prev_end; a[3] += 1
        prev_end = en if prev_end is None else max(prev_end, en)
    print("\n| kernel | grid (threads) | calls | avg us | avg gap before us |")
    print("|---|---|---|---|---|")
    for (name, grid), (n, tot, g, gn) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"| `{name}` | {grid} | {n} | {tot / n / 1e3:.2f} | {(g / gn / 1e3) if gn else 0:.2f} |")


def gaps(c, name_col, thresh_us=30.0, top=25):
    """GPU idle time between consecutive dispatches: total, and the largest gaps with the kernels on either side."""
    q = f"""select d.start, d.end, s.{name_col} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
            on d.kernel_id = s.id order by d.start"""
    rows = list(c.execute(q))
    out, idle_small, idle_big, busy_end = [], 0, 0, rows[0][1]
    for (st, en, name), prev in zip(rows[1:], rows[:-1]):
        g = st - busy_end
        if g > 0:
            if g / 1e3 >= thresh_us:
                idle_big += g
                out.append((g, short(prev[2])[:60], short(name)[:60], (st - rows[0][0]) / 1e6))
            else:
                idle_small += g
        busy_end = max(busy_end, en)
    span = rows[-1][1] - rows[0][0]
    print(f"\nspan {span / 1e6:.2f} ms; idle in gaps < {thresh_us:.0f} us: {idle_small / 1e6:.2f} ms; idle in gaps >= {thresh_us:.0f} us: "
          f"{idle_big / 1e6:.2f} ms ({len(out)} gaps)")
    print("\n| gap us | at ms | after kernel | before kernel |\n|---|---|---|---|")
    for g, a, b, t in sorted(out, reverse=True)[:top]:
        print(f"| {g / 1e3:.1f} | {t:.1f} | `{a}` | `{b}` |")


if __name__ == "__main__":
    main(sys.argv[1])
