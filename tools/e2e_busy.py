"""GPU busy fraction of the end-to-end call's passes from a rocprofv3 kernel trace (rocpd database).

  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/pe -- python bench.py --e2e-only --e2e-slots 1024 --no-cpu-baseline
  python tools/e2e_busy.py /tmp/pe

Dispatch intervals are merged; the trace is cut where the device idles for more than 25 ms (between the leg's passes: host-side page
generation / result checks), and every segment longer than 0.5 s is reported with its busy fraction and its largest idle gaps."""
import glob
import sqlite3
import sys


def main(root):
    db = glob.glob(root + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    rows = c.execute(f"select start, end from {kd} order by start").fetchall()
    names = None
    try:                                                     # kernel names of the dispatches around the largest gaps (schema differs between rocprofv3 builds)
        cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
        ks = [t for t in tabs if "kernel_symbol" in t]
        if ks and "kernel_id" in cols:
            sym = dict(c.execute(f"select id, kernel_name from {ks[0]}").fetchall())
            names = [(s_, e_, sym.get(k_, "?")) for s_, e_, k_ in c.execute(f"select start, end, kernel_id from {kd} order by start")]
    except Exception:
        names = None
    merged = []
    for s, e in rows:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    segs, cur = [], [merged[0]]
    for iv in merged[1:]:
        if iv[0] - cur[-1][1] > 25e6:
            segs.append(cur); cur = [iv]
        else:
            cur.append(iv)
    segs.append(cur)
    for sg in segs:
        span = sg[-1][1] - sg[0][0]
        if span < 0.5e9:
            continue
        busy = sum(e - s for s, e in sg)
        gaps = sorted((sg[i + 1][0] - sg[i][1] for i in range(len(sg) - 1)), reverse=True)
        big = sum(g for g in gaps if g > 50e3)
        print(f"segment {span / 1e6:8.1f} ms: busy {busy / span:.3f}, {len(sg)} merged intervals, idle in gaps > 50 us {big / 1e6:6.1f} ms, "
              f"in gaps 5-50 us {sum(g for g in gaps if 5e3 < g <= 50e3) / 1e6:6.1f} ms, < 5 us {sum(g for g in gaps if g <= 5e3) / 1e6:6.1f} ms; "
              f"largest {[round(g / 1e6, 2) for g in gaps[:6]]} ms")
        where = sorted(((sg[i + 1][0] - sg[i][1], sg[i][1], sg[i + 1][0]) for i in range(len(sg) - 1)), reverse=True)[:8]
        for g, a, b in sorted(where, key=lambda w: w[1]):
            before = after = ""
            if names:
                bb = [n for s_, e_, n in names if e_ == a]
                aa = [n for s_, e_, n in names if s_ == b]
                before, after = (bb[0][:50] if bb else "?"), (aa[0][:50] if aa else "?")
            print(f"      gap {g / 1e6:6.2f} ms at +{(a - sg[0][0]) / 1e6:7.1f} ms   {before}  ->  {after}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pe")
