#!/usr/bin/env python3
"""Time the REAL reference (microsoft/maro built by oracle/build_ref.sh; build container only) on this box and commit the record
bench.py embeds as `cpu_baseline_reference` where the reference is not importable (the GPU box): BASELINE.md section 3, steps 2-3 —
single-process `Env.step` with a random legal agent, and `maro.vector_env.VectorEnv(batch_num=cores)`.

    bash oracle/build_ref.sh && python tools/cpu_reference_baseline.py [seconds per leg = 30]
    -> profiles/cpu_reference_baseline.json
"""
import datetime
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import bench
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    topology, durations = "global_trade.22p_l0.8", 1120
    rec = bench.cpu_baseline_reference(topology, durations, budget)
    if rec is None or "error" in rec:
        raise SystemExit(f"no built reference importable (run oracle/build_ref.sh first): {rec}")
    rec.update({
        "where": "build container (no GPU; the reference cannot travel to the GPU box)",
        "topology": topology, "durations": durations,
        "host_cores": bench.host_cores(), "python": sys.version.split()[0],
        "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
        "head": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip(),
        "produced_by": "tools/cpu_reference_baseline.py (bench.cpu_baseline_reference: the same two legs bench.py runs where the reference is importable)",
    })
    out = os.path.join(REPO, "profiles", "cpu_reference_baseline.json")
    with open(out, "w") as fp:
        json.dump(rec, fp, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
