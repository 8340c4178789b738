#!/usr/bin/env python
"""Per-kernel means of the PMC counter(s) in a rocprofv3 rocpd database (one `--pmc` pass = one database).

    python tools/rocpd_pmc.py gpurun_out/r01d_fetch.db [gpurun_out/r01d_write.db ...] > profiles/r01_d_hbm_traffic.md
    python tools/rocpd_pmc.py --raw some_sq_counters.db          (any counters: mean raw value per kernel, tools/profile_decode_pmc.sh)

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


def bucket(name):
    """bench.py's profiler buckets (gemm_cfg_id in surya_amd/csrc/gemm.h) from the mangled kernel name."""
    if "gemm_nt_p8p_kernel" in name or "gemm_nt_persist_kernel" in name:        # the persistent 256x256 loops (round 5 / round 4)
        return "gemm_nt 128x128 / 256x256 / 256x320 (encoder + prefill GEMMs, lm_head)"
    m = re.search(r"gemm_nt_kernelI..Li(\d+)ELi(\d+)E", name)
    if m:
        bm, bn = int(m.group(1)), int(m.group(2))
        if bm >= 128 and bn >= 128:
            return "gemm_nt 128x128 / 256x256 / 256x320 (encoder + prefill GEMMs, lm_head)"
        return "gemm_nt tall 256x{32,64} (decode-step GEMMs, M<=256)" if bm == 256 else "gemm_nt small tiles"
    return "conv_gemm (implicit-GEMM convolutions, NHWC)" if "conv_gemm_kernel" in name else None


def as_json(rows):
    """{bucket: HBM-side bytes per launch} = (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the bucket / its dispatches."""
    import json
    tot, n = {}, {}
    for name, ctr, cnt, mean, total, dur in rows:
        b = bucket(name)
        if b is None:
            continue
        tot[b] = tot.get(b, 0.0) + total * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        if ctr == "FETCH_SIZE":
            n[b] = n.get(b, 0) + cnt
    print(json.dumps({b: round(tot[b] / n[b]) for b in tot if n.get(b)}, indent=1))


def det_json(rows):
    """{"detection forward (all kernels)": HBM-side bytes per forward} from PMC passes over `bench.py --det-only`: (2 x FETCH_SIZE
    + WRITE_SIZE) summed over every model kernel (post-processing and copies excluded) / number of forwards (= dispatches of the
    final x4 upsample kernel)."""
    import json
    tot, fw = 0.0, 0
    for name, ctr, cnt, mean, total, dur in rows:
        if "post_" in name or "rocclr" in name or "prof_null" in name:
            continue
        tot += total * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        if ctr == "FETCH_SIZE" and "upsample_planes" in name:      # upsample_planes_kernel / upsample_planes4_kernel
            fw += cnt
    print(json.dumps({"detection forward (all kernels)": round(tot / max(fw, 1)), "forwards": fw}, indent=1))


def main(paths):
    js = "--json" in paths
    dj = "--det-json" in paths
    raw = "--raw" in paths
    paths = [p for p in paths if p not in ("--json", "--det-json", "--raw")]
    rows = [r for p in paths for r in load(p)]
    if raw:                     # any counter, no unit interpretation: mean value per dispatch (summed over the counter's instances)
        rows.sort(key=lambda r: (-r[5] * r[2], r[0], r[1]))
        print("| kernel | counter | dispatches | mean value / dispatch | mean us (under PMC) |")
        print("|---|---|---|---|---|")
        for name, ctr, n, mean, tot, dur in rows:
            print(f"| `{short(name)}` | {ctr} | {n} | {mean:.1f} | {dur / 1e3:.2f} |")
        return
    if dj:
        return det_json(rows)
    if js:
        return as_json(rows)
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
