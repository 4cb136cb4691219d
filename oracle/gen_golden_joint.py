#!/usr/bin/env python3
"""Golden vectors for the Joint decision modes (DecisionMode.Joint / JointWithSequentialAction,
maro/simulator/core.py:354-366) from the REAL reference — ORACLE tooling, same recipe as gen_golden.py.

Per yield the reference returns every pending decision event of the tick; recorded per step:
  decisions [V, 8] rows (tick, port, vessel, scope.load, scope.discharge, early_discharge, frame_index, valid) in event
  order, the flat action list that was sent back (one rand0 action per answered event), and n_answered.
Policies: "all" answers every event; "some" answers the first k = rng.randint(1, n) events (in Joint mode the others
are finished without an action, in JointWithSequentialAction they come back — with the payload objects, hence the
action scopes cached at their first read).
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden import MATRIX_ATTRS, PORT_ATTRS, VESSEL_ATTRS  # noqa: E402

CASES = {
    # name: (topology, env kwargs, decision_mode, policy)
    "joint_gt22p_l08_all": ("global_trade.22p_l0.8", dict(durations=120), 1, "all"),
    "joint_gt22p_l08_some": ("global_trade.22p_l0.8", dict(durations=120), 1, "some"),
    "jointseq_gt22p_l08_some": ("global_trade.22p_l0.8", dict(durations=120), 2, "some"),
    "jointseq_toy5p_l05_some": ("toy.5p_ssddd_l0.5", dict(durations=150, snapshot_resolution=3), 2, "some"),
    "joint_toy6p_l08_all": ("toy.6p_sssbdd_l0.8", dict(durations=150), 1, "all"),
}


def worker(maro_root, case_name, out_path):
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    sys.path.insert(0, maro_root)
    import random as pyrandom

    import numpy as np
    from maro.simulator import DecisionMode, Env
    from maro.simulator.scenarios.cim.common import Action, ActionType

    topology, kwargs, mode, policy = CASES[case_name]
    env = Env(scenario="cim", topology=topology, start_tick=0, decision_mode=DecisionMode(mode), **kwargs)
    V = len(env.business_engine._vessels)
    rng = pyrandom.Random(0)
    decs, mets, acts, nans = [], [], [], []
    seen = set()
    m, des, done = env.step(None)
    while not done:
        rows = np.zeros((V, 8), np.int32)
        for i, de in enumerate(des):
            sc = de.action_scope
            rows[i] = [de.tick, de.port_idx, de.vessel_idx, sc.load, sc.discharge, de.early_discharge, env.frame_index, 1]
        k = len(des) if policy == "all" else rng.randint(1, len(des))
        actions, enc = [], -np.ones((V, 4), np.int32)
        ports_here = [d.port_idx for d in des]
        for i, de in enumerate(des[:k]):
            sc = de.action_scope
            # a LOAD is only safe when the cached scope cannot be stale: no other pending vessel at the same port, and
            # the event is shown for the first time (the reference asserts on quantities beyond the live scope)
            fresh = (de.tick, de.vessel_idx) not in seen and ports_here.count(de.port_idx) == 1
            if rng.random() < 0.5 and sc.load > 0 and fresh:
                q, ty = rng.randint(0, sc.load), ActionType.LOAD
            else:
                q, ty = rng.randint(0, sc.discharge), ActionType.DISCHARGE
            actions.append(Action(de.vessel_idx, de.port_idx, q, ty))
            enc[i] = [de.vessel_idx, de.port_idx, q, 0 if ty == ActionType.LOAD else 1]
        seen.update((d.tick, d.vessel_idx) for d in des)
        decs.append(rows); mets.append([m["order_requirements"], m["container_shortage"], m["operation_number"]])
        acts.append(enc); nans.append(k)
        m, des, done = env.step(actions)
    sl = env.snapshot_list
    out = dict(decisions=np.array(decs, np.int32), metrics=np.array(mets, np.int64), actions=np.array(acts, np.int32),
               n_answered=np.array(nans, np.int32),
               final_metrics=np.array([m["order_requirements"], m["container_shortage"], m["operation_number"]], np.int64),
               final_tick=np.array([env.tick], np.int32), frame_indices=np.array(sl.get_frame_index_list(), np.int32),
               snap_ports=sl["ports"][::PORT_ATTRS], snap_vessels=sl["vessels"][::VESSEL_ATTRS],
               snap_matrices=sl["matrices"][::MATRIX_ATTRS],
               meta=np.frombuffer(json.dumps(dict(case=case_name, topology=topology, kwargs=kwargs, decision_mode=mode,
                                                  policy=policy)).encode(), np.uint8))
    np.savez_compressed(out_path, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--case")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        worker(args.maro, args.case, os.path.join(args.out, f"cimjoint_{args.case}.npz"))
        return
    for name in ([args.case] if args.case else CASES):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--maro", args.maro, "--out", args.out, "--case", name,
                               "--worker"])
        print("golden:", name, os.path.getsize(os.path.join(args.out, f"cimjoint_{name}.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
