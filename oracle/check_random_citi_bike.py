#!/usr/bin/env python3
"""Pins the citi_bike oracle against the REAL reference on RANDOM data sets — ORACLE tooling (needs oracle/build_ref.sh and
its stub modules; log committed next to this file).  A random data set of tests/fuzz_citi_bike.py (2..40 stations, random
neighbour graphs, cutting filters, scope ratios, water marks, transfer-time distributions, zero-duration trips) is WRITTEN as a
MARO build folder by maro_amd (data_lib.write_binary: trips.bin, KNYC_daily.bin; csv tables; config.yml), run through the
reference's own Env in a fresh process — Sequential, Joint or JointWithSequentialAction, random start ticks and snapshot
resolutions — and compared with oracle/citi_bike_oracle.py on the natively compiled folder: every decision event, action scope,
metric, and the stations snapshot history.

    python oracle/check_random_citi_bike.py [first_seed=0] [count=40]
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
MARO = os.environ.get("MARO_REFERENCE_BUILD", "/tmp/oracle/maro_src")
STUBS = os.environ.get("MARO_REFERENCE_STUBS", "/tmp/oracle/stubs")


def worker(case_seed):
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    os.environ.setdefault("SKIP_DEPLOYMENT", "TRUE")
    sys.path.insert(0, STUBS)
    sys.path.insert(0, MARO)
    sys.path.insert(0, REPO)   # (the reference checkout has a `tests` package of its own: ours must come first)
    import numpy as np
    import yaml
    from maro.simulator import DecisionMode, Env
    from maro.simulator.scenarios.citi_bike.common import Action, DecisionType

    from maro_amd.citi_bike.data import load_build_folder
    from maro_amd.citi_bike.synthetic import write_build_folder
    from oracle.citi_bike_oracle import STATION_ATTRS, CitiBikeOracle, draw_transfer_times
    from tests.cb_batch_check import policy_action
    from tests.fuzz_citi_bike import random_data
    rng = np.random.RandomState(case_seed)
    raw = random_data(rng)
    kw = dict(durations=int(rng.choice([150, 380])), snapshot_resolution=int(rng.choice([1, 4, 10])))
    if rng.rand() < 0.4:
        kw["max_snapshots"] = int(rng.randint(2, 12))
    if rng.rand() < 0.35 and kw["durations"] == 150:
        kw["start_tick"] = int(rng.choice([7, 33, 101, 240]))
    mode = int(rng.choice([0, 0, 1, 2]))
    folder = tempfile.mkdtemp(prefix="rnd_cb_")
    cfg = write_build_folder(raw, folder, start_utc=1559534400 + 3600 * int(rng.randint(0, 48)), rng=rng)
    with open(os.path.join(folder, "config.yml"), "wt") as fp:
        yaml.safe_dump(cfg, fp)
    data = load_build_folder(cfg, folder, name="rnd", is_holiday=None)   # the reference runs here with the `holidays` stand-in: no holidays
    np.random.seed(case_seed)
    env = Env("citi_bike", folder, **{"start_tick": 0, **kw}, decision_mode=DecisionMode(mode))
    o = CitiBikeOracle(data, transfer_times=draw_transfer_times(data, case_seed, 20000), **kw)
    n = events = 0

    def check_event(de, ode):
        items = [(int(k), int(v)) for k, v in de.action_scope.items()]
        assert (de.tick, de.station_idx, 0 if de.type == DecisionType.Supply else 1, de.frame_index) == (ode["tick"], ode["station_idx"], ode["type"], ode["frame_index"]), (n, de, ode)
        assert items == [tuple(x) for x in ode["action_scope"]], (n, items, ode["action_scope"])

    m, de, done = env.step(None)
    om, ode, odone = o.step(None) if mode == 0 else o.step_joint(None, mode)
    while True:
        assert done == odone, (n, done, odone)
        assert dict(m) == om if not done or m else True, (n, m, om)
        if done:
            break
        n += 1
        if mode == 0:
            check_event(de, ode)
            events += 1
            act = policy_action(n, 0, ode)
            m, de, done = env.step(Action(*act) if act else None)
            om, ode, odone = o.step([act] if act else None)
        else:
            assert len(de) == len(ode), (n, len(de), len(ode))
            acts = []
            for i, (d, od) in enumerate(zip(de, ode)):
                d._action_scope = None          # the scope of the CURRENT state (see oracle/gen_golden_citi_bike_joint.py)
                check_event(d, od)
                acts.append(policy_action(n * 64 + i, 0, od))
            events += len(de)
            k = len(de) if n % 3 == 0 else max(1 if mode == 2 else 0, (n * 7) % (len(de) + 1))
            m, de, done = env.step([Action(*a) if a else None for a in acts[:k]])
            om, ode, odone = o.step_joint([[a] if a else None for a in acts[:k]], mode)
    sl = env.snapshot_list
    assert sl.get_frame_index_list() == o.frame_indices()
    assert np.array_equal(sl["stations"][::STATION_ATTRS], o.query("stations", [], [], STATION_ATTRS))
    assert np.array_equal(sl["matrices"][::"trips_adj"], o.query("matrices", [], [], ["trips_adj"]))
    print(json.dumps(dict(seed=case_seed, steps=n, events=events, mode=mode, stations=data.n_stations, **kw)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
        return
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ok, bad, events = 0, [], 0
    for s in range(first, first + count):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(s)], capture_output=True, text=True, timeout=900)
        if out.returncode == 0:
            line = out.stdout.strip().splitlines()[-1]
            events += json.loads(line)["events"]
            ok += 1
            print(line)
        else:
            bad.append(s)
            print("FAILED seed", s, out.stderr.strip().splitlines()[-1][:300])
    print(f"{ok} of {count} random citi_bike data sets ({events} decision events): the oracle equals the reference (events, scopes, metrics, stations + "
          f"trips_adj snapshot history); failures: {bad}")


if __name__ == "__main__":
    main()
