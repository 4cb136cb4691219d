#!/usr/bin/env python3
"""citi_bike half of tools/gpu_profile.sh: turn the rocprofv3 passes of `python bench.py --scenario citi_bike ...` into one entry of
profiles/latest_pmc_citi_bike.json (HBM bytes per BATCH STEP = all step kernels of one mrx_cb_step call: the wave-cooperative
decision kernel, the general / replay kernel) and a markdown summary.

    python tools/refresh_pmc_citi_bike.py gpurun_out/<tag>/cb_<name> <name> gpurun_out/<tag>
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

STEP_KERNELS = ("mrx_k_cb_step", "mrx_k_cb_step_wave", "mrx_k_cb_replay_wave")


def per_step(path, counter, groups=1):
    """sum over the step kernels of (mean counter value per dispatch x dispatches) / dispatches of the most frequent one, x the env groups
    per GPU (a batch step = one launch of every group)"""
    tot, calls = 0.0, {}
    for name, s, e, gx, wx, lds, ctrs in load_rows(path):
        name = name.replace(".kd", "")
        if name.startswith("mrx_k_cb") and counter in ctrs:      # every kernel of the batch step: step kernels, policy, snapshot query
            tot += ctrs[counter]
            calls[name] = calls.get(name, 0) + 1
    steps = max((v for k, v in calls.items() if k in STEP_KERNELS), default=1)
    return tot / max(steps, 1) * groups, calls


def main():
    folder, name, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    line = json.load(open(os.path.join(folder, "bench_line.json")))
    G = line["config"].get("groups_per_gpu", 1)
    fetch, calls = per_step(db(folder, "fetch"), "FETCH_SIZE", G)
    write, _ = per_step(db(folder, "write"), "WRITE_SIZE", G)
    ent = {"topology": line["metric"].split()[-1], "envs_per_launch": line["config"]["envs_per_gpu"], "step_budget": line["config"].get("step_budget", 0), "replay_period": line["config"].get("replay_period", 1),
           "groups_per_gpu": line["config"].get("groups_per_gpu", 1), "fetch_size_kib": fetch, "write_size_kib": write, "kernels": calls, "bench_value": line["value"], "bench_ms_per_step": line["ms_per_step"],
           "specialized_kernels": line["config"]["specialized_kernels"], "code_object_key": line["config"].get("code_object_key"), "git_head": os.environ.get("GIT_HEAD")}
    path = os.path.join(out_dir, "latest_pmc_citi_bike.json")
    rec = json.load(open(path)) if os.path.exists(path) else {
        "source": f"profiles/{os.path.basename(os.path.normpath(out_dir))}_citi_bike.md (tools/gpu_profile.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction; bytes per batch step = every mrx_k_cb_* kernel of one batch step: policy, step kernels, snapshot query)",
        "kernel": "every kernel of one batch step (mrx_k_cb_random_policy, mrx_k_cb_step | mrx_k_cb_step_wave + mrx_k_cb_replay_wave, mrx_k_cb_query*)", "entries": []}
    rec["entries"] = [x for x in rec["entries"] if not (x["topology"] == ent["topology"] and x["envs_per_launch"] == ent["envs_per_launch"] and x.get("step_budget", 0) == ent["step_budget"] and x.get("replay_period", 1) == ent["replay_period"] and x.get("groups_per_gpu", 1) == ent["groups_per_gpu"])] + [ent]
    json.dump(rec, open(path, "w"), indent=1)
    buf = io.StringIO()
    with redirect_stdout(buf):
        print(f"# citi_bike {name}: bench line {line['value'] / 1e6:.2f} M env-steps/s, {line['ms_per_step']:.4f} ms per batch step ({line['config']['workload']})\n")
        print(f"HBM bytes per batch step (all step kernels): 2 x FETCH_SIZE {fetch:.1f} KiB + WRITE_SIZE {write:.1f} KiB = {(2 * fetch + write) * 1024 / 1e6:.2f} MB\n")
        for p in ("trace", "fetch", "write", "sq"):
            d = db(folder, p)
            if d:
                print(f"## {p} pass\n")
                summarise(d)
                export_dispatches(d, os.path.join(folder, f"{p}_dispatches.csv.gz"))
    with open(os.path.join(out_dir, os.path.basename(os.path.normpath(out_dir)) + "_citi_bike.md"), "a") as fp:
        fp.write(buf.getvalue() + "\n")
    print(json.dumps(ent))


if __name__ == "__main__":
    main()
