#!/usr/bin/env python3
"""Turn the rocprofv3 passes of `python bench.py` into profiles/latest_pmc.json (what bench.py reports as roofline.traffic)
and a markdown summary.  Run the passes on the GPU box (tools/gpu_profile.sh does), then this script on the result folder:

    python tools/refresh_pmc.py gpurun_out/<tag> profiles/<name>.md [out_dir]

(out_dir given: latest_pmc.json and the summary go there instead of profiles/ — the GPU box only returns gpurun_out/, and
the raw databases are too large to come back: tools/gpu_profile.sh summarises on the box and deletes them.)

Expects <folder>/{trace,fetch,write[,sq]}/**/r_results.db (one PMC counter set per pass, --kernel-trace only) and
<folder>/bench_line.json (the un-profiled bench line of the same run)."""
import glob
import io
import json
import os
import sqlite3
import sys
from contextlib import redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.rocprof_summary import summarise  # noqa: E402


def db(folder, name):
    hits = glob.glob(os.path.join(folder, name, "**", "r_results.db"), recursive=True)
    return hits[0] if hits else None


def mean_counter(path, kernel, counter):
    c = sqlite3.connect(path)
    names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    pmc_names = dict(c.execute("select id, name from rocpd_info_pmc"))
    ev_kernel = {ev: names.get(kid, "").replace(".kd", "") for kid, ev in c.execute("select kernel_id, event_id from rocpd_kernel_dispatch")}
    tot = n = 0
    for ev, pid, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
        if pmc_names.get(pid) == counter and ev_kernel.get(ev) == kernel:
            tot += val
            n += 1
    return tot / max(n, 1), n


def export_dispatches(path, out_csv):
    """One CSV row per kernel dispatch (name, start, end, grid, workgroup, LDS, counters) — the raw material of the tables, small
    enough to come back from the GPU box; tools/rocprof_summary.py re-derives the .md from it (`summarise_csv`)."""
    import csv
    import gzip
    c = sqlite3.connect(path)
    names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    pmc_names = dict(c.execute("select id, name from rocpd_info_pmc")) if c.execute("select count(*) from rocpd_info_pmc").fetchone()[0] else {}
    pmc = {}
    for ev, pid, val in c.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
        pmc.setdefault(ev, {})[pmc_names.get(pid, str(pid))] = val
    cols = sorted({k for d in pmc.values() for k in d})
    with gzip.open(out_csv, "wt", newline="") as fp:
        w = csv.writer(fp)
        w.writerow(["kernel", "start_ns", "end_ns", "grid_x", "workgroup_x", "lds_bytes"] + cols)
        for kid, ev, st, en, gx, wx, lds in c.execute("select kernel_id, event_id, start, end, grid_size_x, workgroup_size_x, group_segment_size from rocpd_kernel_dispatch"):
            w.writerow([names.get(kid, str(kid)).replace(".kd", ""), st, en, gx, wx, lds] + [pmc.get(ev, {}).get(k, "") for k in cols])


def main():
    folder, out_md = sys.argv[1], sys.argv[2]
    out_dir = sys.argv[3] if len(sys.argv) > 3 else None
    line = json.load(open(os.path.join(folder, "bench_line.json")))
    kernel = line["roofline"]["kernel"]
    fetch, nf = mean_counter(db(folder, "fetch"), kernel, "FETCH_SIZE")
    write, nw = mean_counter(db(folder, "write"), kernel, "WRITE_SIZE")
    rec = {"source": out_md, "kernel": kernel, "envs_per_launch": line["config"]["envs_per_launch"], "topology": line["config"]["workload"].split(",")[0].split()[-1],
           "fetch_size_kib": fetch, "write_size_kib": write, "dispatches": [nf, nw], "step_mode": line["config"]["step_mode"],
           "mean_tick_at_window_start": line["config"].get("mean_tick_at_window_start"),
           # what was profiled: the step kernels' cache key / image hash (bench.py only uses this record for the SAME code object),
           # the commit the GPU run was started from (GIT_HEAD is passed in by the caller: the box has no .git), the run's own value
           "code_object_key": line["config"].get("code_object_key"), "code_object_sha16": line["config"].get("code_object_sha16"),
           "git_head": os.environ.get("GIT_HEAD"), "bench_value": line["value"], "bench_ms_per_step": line["ms_per_step"],
           "date": __import__("datetime").datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --steps 60 --warmup 50 --no-cpu --no-episode --parity-envs 0` "
                   "(mid-episode after the preroll); FETCH_SIZE doubled by the reader (gfx950 correction, MI355X_MICROARCH.md HBM section)"}
    with open(os.path.join(out_dir, "latest_pmc.json") if out_dir else os.path.join(REPO, "profiles", "latest_pmc.json"), "w") as fp:
        json.dump(rec, fp, indent=1)
    buf = io.StringIO()
    with redirect_stdout(buf):
        for name in ("trace", "fetch", "write", "sq"):
            p = db(folder, name)
            if p:
                print(f"## {name} pass\n")
                summarise(p)
                export_dispatches(p, os.path.join(folder, f"{name}_dispatches.csv.gz"))   # kept under gpurun_out/<tag>/ (the .db is not)
    traffic = (2 * fetch + write) * 1024
    head = (f"git HEAD {os.environ.get('GIT_HEAD')}, step-kernel code object {rec['code_object_key']} (sha {rec['code_object_sha16']}); raw per-dispatch rows: "
            f"gpurun_out/{os.path.basename(os.path.normpath(folder))}/*_dispatches.csv.gz (`python tools/rocprof_summary.py <file>.csv.gz` re-derives the tables)\n\n"
            f"bench line of this run (un-profiled): value {line['value'] / 1e6:.1f} M env-steps/s, ms_per_step {line['ms_per_step']:.4f}, "
            f"end to end {line.get('value_end_to_end', 0) / 1e6:.1f} M; parity {line.get('parity', {}).get('ok')} on {line.get('parity', {}).get('envs_checked')} envs\n\n"
            f"dominant kernel `{kernel}`: FETCH_SIZE {fetch:.1f} KiB (x2) + WRITE_SIZE {write:.1f} KiB = {traffic / 1e6:.1f} MB per launch of "
            f"{line['config']['envs_per_launch']:.0f} env-steps = {traffic / line['config']['envs_per_launch'] / 1e3:.1f} KB per env-step\n\n")
    with open(os.path.join(out_dir, os.path.basename(out_md)) if out_dir else os.path.join(REPO, out_md), "w") as fp:
        fp.write(head + buf.getvalue())
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
