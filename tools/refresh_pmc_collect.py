#!/usr/bin/env python3
"""Config 5 half of tools/gpu_profile.sh: turn the rocprofv3 passes of `python bench.py --policy dqn --collect ...` into one entry of
profiles/latest_pmc_collect.json — HBM bytes per BATCH INTERACTION of the collection loop = (sum over EVERY kernel the loop launches:
DQN bin / forward, record, step, schedule, emit, the sampler's few tensor ops; reset / order-table kernels excluded) / interactions,
where interactions = dispatches of the step kernel / groups — and a markdown summary with the per-kernel table.

    python tools/refresh_pmc_collect.py gpurun_out/<tag>/collect_<envs> <envs> gpurun_out/<tag>
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.refresh_pmc import db, export_dispatches  # noqa: E402
from tools.rocprof_summary import load_rows, summarise  # noqa: E402

EXCLUDED = ("mrx_k_cim_reset", "mrx_k_cim_order_table")


def per_interaction(path, counter, groups):
    tot, calls, by_kernel = 0.0, {}, {}
    for name, s, e, gx, wx, lds, ctrs in load_rows(path):
        name = name.replace(".kd", "")
        if counter not in ctrs or name.startswith(EXCLUDED):
            continue
        tot += ctrs[counter]
        calls[name] = calls.get(name, 0) + 1
        by_kernel[name] = by_kernel.get(name, 0.0) + ctrs[counter]
    steps = sum(v for k, v in calls.items() if k.startswith("mrx_k_cim_step"))
    inter = max(steps / max(groups, 1), 1.0)
    return tot / inter, inter, {k: v / inter for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:12]}


def main():
    folder, name, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    line = json.load(open(os.path.join(folder, "bench_line.json")))
    G = line["config"]["groups_per_gpu"]
    fetch, inter, f_by = per_interaction(db(folder, "fetch"), "FETCH_SIZE", G)
    write, _, w_by = per_interaction(db(folder, "write"), "WRITE_SIZE", G)
    ent = {"envs_per_gpu": line["config"]["envs_per_gpu"], "groups_per_gpu": G, "fetch_size_kib": fetch, "write_size_kib": write, "interactions_profiled": inter,
           "fetch_kib_by_kernel": f_by, "write_kib_by_kernel": w_by, "code_object_key": line["config"].get("code_object_key"),
           "bench_value": line["value"], "bench_ms_per_step": line["ms_per_step"], "git_head": os.environ.get("GIT_HEAD")}
    path = os.path.join(out_dir, "latest_pmc_collect.json")
    rec = json.load(open(path)) if os.path.exists(path) else {
        "source": f"profiles/{os.path.basename(os.path.normpath(out_dir))}_collect.md (tools/gpu_profile.sh COLLECT=1: separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled "
                  "per the gfx950 correction; bytes per batch interaction = every kernel of the loop except reset / order table)", "entries": []}
    rec["entries"] = [x for x in rec["entries"] if not (x["envs_per_gpu"] == ent["envs_per_gpu"] and x["groups_per_gpu"] == G)] + [ent]
    json.dump(rec, open(path, "w"), indent=1)
    buf = io.StringIO()
    with redirect_stdout(buf):
        print(f"# collection loop, {name} envs per GPU: bench line {line['value'] / 1e6:.2f} M env-steps/s, {line['ms_per_step']:.4f} ms per batch interaction\n")
        print(f"HBM bytes per batch interaction (every kernel of the loop): 2 x FETCH_SIZE {fetch:.1f} KiB + WRITE_SIZE {write:.1f} KiB = {(2 * fetch + write) * 1024 / 1e6:.2f} MB "
              f"over {inter:.0f} profiled interactions; roofline_policy of the line: {json.dumps(line.get('roofline_policy'))}\n")
        for p in ("trace", "fetch", "write"):
            d = db(folder, p)
            if d:
                print(f"## {p} pass\n")
                summarise(d)
                export_dispatches(d, os.path.join(folder, f"{p}_dispatches.csv.gz"))
    with open(os.path.join(out_dir, os.path.basename(os.path.normpath(out_dir)) + "_collect.md"), "a") as fp:
        fp.write(buf.getvalue() + "\n")
    print(json.dumps(ent))


if __name__ == "__main__":
    main()
