#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (`*_results.db`) into the small text tables kept in profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/trace/r1_results.db [more.db ...] > profiles/rNN_xxx.md

Per kernel: dispatch count, total / mean / min / max duration; and, when the run collected PMC
counters, the mean counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in
KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM), so the
table prints both the raw and the corrected (x2) read bytes.
"""
import sqlite3
import sys


def load_rows(path):
    """(kernel, start, end, grid, wg, lds, {counter: value}) per dispatch, from a rocpd database or from the per-dispatch CSV
    export tools/refresh_pmc.py leaves under gpurun_out/<tag>/ (`*_dispatches.csv.gz`)."""
    if path.endswith(".csv") or path.endswith(".csv.gz"):
        import csv
        import gzip
        with (gzip.open(path, "rt", newline="") if path.endswith(".gz") else open(path, newline="")) as fp:
            rd = csv.reader(fp)
            head = next(rd)
            out = []
            for r in rd:
                out.append((r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), {k: float(v) for k, v in zip(head[6:], r[6:]) if v != ""}))
            return out
    c = sqlite3.connect(path)
    names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    rows = c.execute("select kernel_id, event_id, start, end, grid_size_x, workgroup_size_x, group_segment_size "
                     "from rocpd_kernel_dispatch").fetchall()
    pmc_names = dict(c.execute("select id, name from rocpd_info_pmc")) if c.execute(
        "select count(*) from rocpd_info_pmc").fetchone()[0] else {}
    pmc = {}
    for ev, pid, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
        pmc.setdefault(ev, {})[pmc_names.get(pid, str(pid))] = val
    return [(names.get(kid, str(kid)), s, e, gx, wx, lds, pmc.get(ev, {})) for kid, ev, s, e, gx, wx, lds in rows]


def summarise(path):
    rows = load_rows(path)
    agg = {}
    for name, s, e, gx, wx, lds, ctrs in rows:
        a = agg.setdefault(name, dict(n=0, tot=0, mn=1 << 62, mx=0, grid=gx, wg=wx, lds=lds, pmc={}))
        d = e - s
        a["n"] += 1
        a["tot"] += d
        a["mn"] = min(a["mn"], d)
        a["mx"] = max(a["mx"], d)
        for k, v in ctrs.items():
            a["pmc"][k] = a["pmc"].get(k, 0.0) + v
    # overlap of dispatches of the same kernel (several streams): sum of durations / union of their intervals
    spans = {}
    for name, st, en, gx, wx, lds, ctrs in rows:
        spans.setdefault(name, []).append((st, en))
    for name, iv in spans.items():
        iv.sort()
        union, cs, ce = 0, iv[0][0], iv[0][1]
        for st, en in iv[1:]:
            if st > ce:
                union += ce - cs
                cs, ce = st, en
            else:
                ce = max(ce, en)
        union += ce - cs
        agg[name]["in_flight"] = agg[name]["tot"] / max(union, 1)
    total = sum(a["tot"] for a in agg.values()) or 1
    print(f"### {path}\n")
    print("| kernel | calls | total ms | mean us | min us | max us | % | grid | wg | LDS B | counters (mean per dispatch) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        ctr = []
        for k, v in a["pmc"].items():
            m = v / a["n"]
            if k == "FETCH_SIZE":
                ctr.append(f"FETCH_SIZE {m:.1f} KiB (x2 gfx950 correction: {2 * m * 1024 / 1e6:.2f} MB)")
            elif k == "WRITE_SIZE":
                ctr.append(f"WRITE_SIZE {m:.1f} KiB ({m * 1024 / 1e6:.2f} MB)")
            else:
                ctr.append(f"{k} {m:.4g}")
        if a.get("in_flight", 1.0) > 1.05:
            ctr.append("in flight %.2f" % a["in_flight"])
        print(f"| {name.replace('.kd', '')} | {a['n']} | {a['tot'] / 1e6:.3f} | {a['tot'] / a['n'] / 1e3:.1f} | "
              f"{a['mn'] / 1e3:.1f} | {a['mx'] / 1e3:.1f} | {100 * a['tot'] / total:.1f} | {a['grid']} | {a['wg']} | "
              f"{a['lds']} | {'; '.join(ctr)} |")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
