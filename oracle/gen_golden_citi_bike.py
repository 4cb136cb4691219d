#!/usr/bin/env python3
"""Golden vectors for the citi_bike path from the REAL reference — ORACLE tooling.

Needs the reference built as for gen_golden.py, plus two stub modules on sys.path (the real packages are not
installed here): `holidays` (US() that contains nothing) and `geopy.distance` (haversine) — see SURVEY.md §8c.
The toy data (trips.bin etc.) was generated ONCE by the reference's own pipeline
(`CitiBikeProcess(is_temp=False).topologies[name].download/clean/build`, unseeded random) and compiled with
tools/import_maro_citi_bike.py into the packaged .npz; on a fresh reference build the folders and config.yml variants are written
back from those .npz files (oracle/setup_toy_topologies.py, which also proves that all 15 toy goldens regenerate byte for byte).

    python oracle/gen_golden_citi_bike.py --maro /tmp/oracle/maro_src --stubs /tmp/oracle/stubs --out tests/golden
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
STATION_ATTRS = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "id", "weekday", "temperature",
                 "weather", "holiday", "extra_cost", "transfer_cost", "failed_return", "min_bikes"]
CASES = {
    "cb_toy3s4t_d1440_r10": ("toy.3s_4t", dict(durations=1440, snapshot_resolution=10), "half"),
    "cb_toy3s4t_d600_r1": ("toy.3s_4t", dict(durations=600, snapshot_resolution=1), "half"),
    "cb_toy3s4t_d2000_r7_ring12": ("toy.3s_4t", dict(durations=2000, snapshot_resolution=7, max_snapshots=12), "all"),
    "cb_toy3s4t_d900_r10_none": ("toy.3s_4t", dict(durations=900, snapshot_resolution=10), "none"),
    # toy.3s_tight: toy.3s_4t trips with small, nearly full stations (overflow -> move_to_neighbor, failed returns,
    # Supply decisions), extra_cost_mode target, transfer time N(3, 2) (zero / negative transfer times)
    "cb_tight_d1500_r10_all": ("toy.3s_tight", dict(durations=1500, snapshot_resolution=10), "all"),
    "cb_tight_d700_r3_half": ("toy.3s_tight", dict(durations=700, snapshot_resolution=3), "half"),
    # the reference's larger toy topologies (4 and 5 stations: more neighbours for the filters to choose from)
    "cb_toy4s4t_d1500_r10_all": ("toy.4s_4t", dict(durations=1500, snapshot_resolution=10), "all"),
    "cb_toy5s6t_d1200_r5_half": ("toy.5s_6t", dict(durations=1200, snapshot_resolution=5, max_snapshots=40), "half"),
    "cb_toy5s6t_d2000_r10_all": ("toy.5s_6t", dict(durations=2000, snapshot_resolution=10), "all"),
    # toy.5s_6t data with filters that really cut the neighbour list (distance 3 -> requirements 2 -> trip window 1 of 4
    # frames), scope ratios 0.2 / 0.7, water marks 0.7 / 0.3; and a `windows: 0` trip-window filter (Python's lst[-0:])
    "cb_filters_d1600_r10_all": ("toy.5s_filters", dict(durations=1600, snapshot_resolution=10), "all"),
    "cb_filters_d1000_r3_half": ("toy.5s_filters", dict(durations=1000, snapshot_resolution=3, max_snapshots=7), "half"),
    "cb_win0_d1400_r10_all": ("toy.5s_win0", dict(durations=1400, snapshot_resolution=10), "all"),
    "cb_tight_d800_r20_none": ("toy.3s_tight", dict(durations=800, snapshot_resolution=20, max_snapshots=6), "none"),
    # city.180s: the seeded synthetic city-shaped topology (maro_amd/citi_bike/synthetic.py: 180 stations, 24 neighbours each,
    # the default distance 20 -> requirements 10 -> trip_window 6 chain), written as a MARO build folder by maro_amd's own
    # binary writer and run through the reference's Env: ~60 deciding stations per decision tick
    # start_tick not a multiple of the snapshot resolution: a frame's post_step snapshot falls on its FIRST ticks and the
    # pre-decision snapshots (core.py:345) of later ticks of the frame are what stays in the snapshot list
    "cb_filters_start27_d500_r7_ring5_half": ("toy.5s_filters", dict(start_tick=27, durations=500, snapshot_resolution=7, max_snapshots=5), "half"),
    "cb_tight_start13_d600_r10_all": ("toy.3s_tight", dict(start_tick=13, durations=600, snapshot_resolution=10), "all"),
    "cb_city180_d1440_r10_ring16_half": ("city.180s", dict(durations=1440, snapshot_resolution=10, max_snapshots=16), "half"),
    "cb_city180_d2000_r20_ring9_all": ("city.180s", dict(durations=2000, snapshot_resolution=20, max_snapshots=9), "all"),
    # city.800s: the SIZE of the reference's shipped ny.* topologies (800 stations, 100 neighbours each, the ny filter chain
    # distance 80 -> requirements 40 -> trip_window 20 over 10 windows, a month of 1-minute ticks, 1.5 M trips); two windows of
    # the month (the reference's pure-Python Env needs minutes per simulated day at this size)
    "cb_city800_d300_r20_ring12_half": ("city.800s", dict(durations=300, snapshot_resolution=20, max_snapshots=12), "half"),
    "cb_city800_start20160_d260_r10_ring3_all": ("city.800s", dict(start_tick=20160, durations=260, snapshot_resolution=10, max_snapshots=3), "all"),
}


def ensure_synthetic(maro_root, topology):
    """city.*: (re)generate the build folder + config.yml and register the topology with the reference checkout; toy.*: write them
    back from the packaged data (oracle/setup_toy_topologies.py) — so every case runs on a fresh `oracle/build_ref.sh` build."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    import yaml
    from maro_amd.citi_bike.synthetic import PACKAGED, build_packaged
    home = os.environ.get("MARO_ORACLE_HOME", "/tmp/oracle/home")
    if topology not in PACKAGED:
        # toy.*: the build folder + config.yml written back from the packaged .npz when a fresh reference build lacks them
        from setup_toy_topologies import ensure_toy
        ensure_toy(maro_root, home, topology)
        return
    bd = os.path.join(home, ".maro", "data", "citi_bike", ".build", topology)
    cfg, _ = build_packaged(topology, bd)
    tdir = os.path.join(maro_root, "maro", "simulator", "scenarios", "citi_bike", "topologies", topology)
    os.makedirs(tdir, exist_ok=True)
    with open(os.path.join(tdir, "config.yml"), "wt") as fp:
        yaml.safe_dump(cfg, fp)


def worker(maro_root, stubs, case, out_path):
    os.environ["HOME"] = os.environ.get("MARO_ORACLE_HOME", "/tmp/oracle/home")  # where the built toy data lives (~/.maro)
    sys.path.insert(0, stubs)
    sys.path.insert(0, maro_root)
    import numpy as np
    from maro.simulator import Env
    from maro.simulator.scenarios.citi_bike.common import Action, DecisionType

    topology, kwargs, policy = CASES[case]
    ensure_synthetic(maro_root, topology)
    np.random.seed(0)
    env = Env("citi_bike", topology, **{"start_tick": 0, **kwargs})
    decs, scopes, mets, acts = [], [], [], []
    m, de, done = env.step(None)
    while not done:
        scope = de.action_scope                      # reading it is what updates the trip-window cache
        items = [(int(k), int(v)) for k, v in scope.items()]
        decs.append([de.tick, de.station_idx, 0 if de.type == DecisionType.Supply else 1, de.frame_index, len(items)])
        scopes.append(items)
        mets.append([m["trip_requirements"], m["bike_shortage"], m["operation_number"]])
        action, enc = None, (-1, -1, -1)
        others = [k for k, _ in items if k != de.station_idx]
        if policy != "none" and others:
            to = others[0]
            n = min(scope[de.station_idx], scope[to])
            n = n // 2 if policy == "half" else n
            frm, dst = (de.station_idx, to) if de.type == DecisionType.Supply else (to, de.station_idx)
            action, enc = Action(frm, dst, int(n)), (frm, dst, int(n))
        acts.append(enc)
        m, de, done = env.step(action)
    sl = env.snapshot_list
    width = max([8] + [len(x) for x in scopes])   # (8 for every toy; the ny filter chain keeps 20 neighbours + the station itself)
    scopes = [x + [(-1, -1)] * (width - len(x)) for x in scopes]
    out = dict(decisions=np.array(decs, np.int32).reshape(-1, 5), scopes=np.array(scopes, np.int32).reshape(-1, width, 2),
               metrics=np.array(mets, np.int64).reshape(-1, 3), actions=np.array(acts, np.int32).reshape(-1, 3),
               final_metrics=np.array([m["trip_requirements"], m["bike_shortage"], m["operation_number"]], np.int64),
               frame_indices=np.array(sl.get_frame_index_list(), np.int32),
               snap_stations=sl["stations"][::STATION_ATTRS], snap_matrices=sl["matrices"][::"trips_adj"],
               final_tick=np.array([env.tick], np.int32),
               meta=np.frombuffer(json.dumps(dict(case=case, topology=topology, kwargs=kwargs, policy=policy, np_seed=0)).encode(), np.uint8))
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
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--maro", a.maro, "--stubs", a.stubs, "--out", a.out,
                               "--case", name, "--worker"])
        print("golden:", name, os.path.getsize(os.path.join(a.out, f"{name}.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
