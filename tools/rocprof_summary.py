#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/gpu_profile.sh into a markdown table.

    python tools/rocprof_summary.py gpurun_out/prof_r01 profiles/r01_rocprof_summary.md

kernel-trace pass -> calls / average duration per kernel; PMC passes -> FETCH_SIZE / WRITE_SIZE (KB per
dispatch).  Bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read (MI355X_MICROARCH.md, section HBM), WRITE_SIZE is uncalibrated.  Both derive from
the L2's memory-side (fabric) request counters: they are L2 <-> fabric bytes -- Infinity-Cache hits are in them -- an upper
bound on what reaches the HBM stacks, which is how the column is named.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"rocprim::(?:wrapped_)?(\w+)_config", name)
    if "rocprim" in name and m:
        return "rocprim::" + m.group(1)
    name = name.replace("rgcn::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name").fetchall()
    out = {}
    for name, n, tot, avg, mn, mx in rows:
        s = short(name)
        a = out.setdefault(s, [0, 0.0, 1e30, 0.0])
        a[0] += n; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    return out


def counter_avg(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), sum(value) from counters_collection "
                     "where counter_name=? group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, tot in rows:
        s = short(name)
        a = out.setdefault(s, [0, 0.0])
        a[0] += n; a[1] += tot
    return {k: v[1] / max(v[0], 1) for k, v in out.items()}


def main():
    d, outp = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-kernel-profile"
    ks = kernel_stats(d + "/trace/trace_results.db")
    try:
        fetch = counter_avg(d + "/pmc_fetch/fetch_results.db", "FETCH_SIZE")
        write = counter_avg(d + "/pmc_write/write_results.db", "WRITE_SIZE")
    except Exception as e:  # noqa: BLE001
        print("no PMC data:", e)
        fetch, write = {}, {}
    total = sum(v[1] for v in ks.values())
    lines = ["# rocprofv3 summary (%s)" % d, "",
             "command: `%s` " % cmd +
             "(25 steps incl. warm-up); kernel-trace pass for durations, separate `--pmc FETCH_SIZE` / "
             "`--pmc WRITE_SIZE` passes for bytes.", "",
             "L2<->fabric MB/launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / 1e6: bytes between the XCDs' L2s and the fabric, Infinity-Cache hits included (gfx950 FETCH_SIZE half-count "
             "correction; WRITE_SIZE uncalibrated).", "",
             "| kernel | calls | avg us | min us | max us | % of GPU time | FETCH_SIZE KB | WRITE_SIZE KB | L2<->fabric MB/launch |",
             "|---|---|---|---|---|---|---|---|---|"]
    for k, (n, tot, mn, mx) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        f, w = fetch.get(k), write.get(k)
        hbm = "" if f is None or w is None else "%.2f" % ((2 * f + w) * 1024 / 1e6)
        lines.append("| %s | %d | %.2f | %.2f | %.2f | %.1f | %s | %s | %s |" % (
            k, n, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total,
            "" if f is None else "%.1f" % f, "" if w is None else "%.1f" % w, hbm))
    lines.append("")
    lines.append("total GPU kernel time: %.3f ms over the run" % (total / 1e6))
    open(outp, "w").write("\n".join(lines) + "\n")
    # machine-readable HBM traffic per launch (bytes) for bench.py's roofline.traffic
    import json
    traffic = {k: (2 * fetch[k] + write[k]) * 1024 for k in ks if k in fetch and k in write}
    json.dump({"source": d, "unit": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024",
               "kernels": traffic}, open(outp.replace(".md", "_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
