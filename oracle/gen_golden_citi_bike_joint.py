#!/usr/bin/env python3
"""Golden vectors for citi_bike in the Joint decision modes (core.py:354-366) from the REAL reference — ORACLE tooling.
Setup as oracle/gen_golden_citi_bike.py (oracle/build_ref.sh + the packaged toy build folders).

Every step the policy reads the scope of EVERY reported event (after clearing the payload's cached scope, so a re-reported
event of JointWithSequentialAction is recorded with the scope of the CURRENT state — what the C ABI reports; the reference
object re-serves its first read, which the object API reproduces on top), answers the first `n_answered` events
(`answer` = "all" | "alt": all / all-but-the-last on alternating steps | "one": only the first) with "half of what fits"
actions, and leaves the rest to the mode: finished (Joint) or reported again (JointWithSequentialAction).

    python oracle/gen_golden_citi_bike_joint.py [--case NAME]
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden_citi_bike import STATION_ATTRS, ensure_synthetic  # noqa: E402

CASES = {
    "cbjoint_toy5s6t_m1_all": ("toy.5s_6t", dict(durations=900, snapshot_resolution=10), 1, "all"),
    "cbjoint_toy5s6t_m1_alt": ("toy.5s_6t", dict(durations=900, snapshot_resolution=5, max_snapshots=30), 1, "alt"),
    "cbjoint_toy5s6t_m2_one": ("toy.5s_6t", dict(durations=700, snapshot_resolution=10), 2, "one"),
    "cbjoint_tight_m1_all": ("toy.3s_tight", dict(durations=1200, snapshot_resolution=10), 1, "all"),      # zero / negative transfer times
    "cbjoint_tight_m2_alt": ("toy.3s_tight", dict(durations=900, snapshot_resolution=3), 2, "alt"),
    "cbjoint_filters_m2_alt": ("toy.5s_filters", dict(durations=1000, snapshot_resolution=10), 2, "alt"),
    "cbjoint_city180_m1_alt": ("city.180s", dict(durations=600, snapshot_resolution=10, max_snapshots=8), 1, "alt"),
}
SCOPE_CAP = 8


def worker(maro_root, stubs, case, out_path):
    os.environ["HOME"] = os.environ.get("MARO_ORACLE_HOME", "/tmp/oracle/home")
    sys.path.insert(0, stubs)
    sys.path.insert(0, maro_root)
    import numpy as np
    from maro.simulator import DecisionMode, Env
    from maro.simulator.scenarios.citi_bike.common import Action, DecisionType

    topology, kwargs, mode, answer = CASES[case]
    ensure_synthetic(maro_root, topology)
    np.random.seed(0)
    env = Env("citi_bike", topology, start_tick=0, decision_mode=DecisionMode(mode), **kwargs)
    n_ev, n_ans, decs, scopes, acts, mets = [], [], [], [], [], []
    m, des, done = env.step(None)
    step = 0
    while not done:
        assert isinstance(des, list) and des
        actions = []
        for de in des:
            de._action_scope = None                      # see the module docstring
            scope = de.action_scope
            items = [(int(k), int(v)) for k, v in scope.items()]
            decs.append([de.tick, de.station_idx, 0 if de.type == DecisionType.Supply else 1, de.frame_index, len(items)])
            scopes.append(items + [(-1, -1)] * (SCOPE_CAP - len(items)))
            others = [k for k, _ in items if k != de.station_idx]
            enc, a = (-1, -1, -1), None
            if others:
                to = others[0]
                n = min(scope[de.station_idx], scope[to]) // 2
                frm, dst = (de.station_idx, to) if de.type == DecisionType.Supply else (to, de.station_idx)
                a, enc = Action(frm, dst, int(n)), (frm, dst, int(n))
            actions.append(a)
            acts.append(enc)
        k = len(des) if answer == "all" else 1 if answer == "one" else (len(des) if step % 2 == 0 else max(len(des) - 1, 0))
        n_ev.append(len(des))
        n_ans.append(k)
        mets.append([m["trip_requirements"], m["bike_shortage"], m["operation_number"]])
        m, des, done = env.step(actions[:k])
        step += 1
    sl = env.snapshot_list
    out = dict(n_events=np.array(n_ev, np.int32), n_answered=np.array(n_ans, np.int32), decisions=np.array(decs, np.int32).reshape(-1, 5),
               scopes=np.array(scopes, np.int32).reshape(-1, SCOPE_CAP, 2), actions=np.array(acts, np.int32).reshape(-1, 3),
               metrics=np.array(mets, np.int64).reshape(-1, 3),
               final_metrics=np.array([m["trip_requirements"], m["bike_shortage"], m["operation_number"]], np.int64),
               frame_indices=np.array(sl.get_frame_index_list(), np.int32), snap_stations=sl["stations"][::STATION_ATTRS],
               final_tick=np.array([env.tick], np.int32),
               meta=np.frombuffer(json.dumps(dict(case=case, topology=topology, kwargs=kwargs, decision_mode=mode, answer=answer, np_seed=0)).encode(), np.uint8))
    np.savez_compressed(out_path, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--stubs", default="/tmp/oracle/stubs")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--case")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        worker(a.maro, a.stubs, a.case, os.path.join(a.out, f"{a.case}.npz"))
        return
    for name in ([a.case] if a.case else CASES):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--maro", a.maro, "--stubs", a.stubs, "--out", a.out, "--case", name, "--worker"])
        print("golden:", name, os.path.getsize(os.path.join(a.out, f"{name}.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
