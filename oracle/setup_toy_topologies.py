#!/usr/bin/env python3
"""ORACLE TOOLING — write the citi_bike TOY build folders + config.yml variants back from the packaged .npz files, so that every
citi_bike golden regenerates from a fresh `oracle/build_ref.sh` alone.

The toy data (`toy.3s_4t`, `toy.4s_4t`, `toy.5s_6t`) was generated ONCE by the reference's own toy pipeline
(`maro/cli/data_pipeline/citi_bike.py` CitiBikeToyPipeline: unseeded random trips, so a second run gives other data); the
variants `toy.3s_tight`, `toy.5s_filters`, `toy.5s_win0` are the same trips with another `decision` block.  What the reference's
business engine reads from such a folder (business_engine.py:205-260) is exactly what the packaged form holds: trips per tick
in file order, stations, the distance table, per-day calendar features.  This script writes that back through the native
writer (`maro_amd.citi_bike.synthetic.write_build_folder` -> `maro_amd.data_lib.write_binary`) into
`<home>/.maro/data/citi_bike/.build/<name>/` and registers `<maro>/maro/simulator/scenarios/citi_bike/topologies/<name>/config.yml`
with the reference checkout; it then reads the folder back with the native loader and asserts that every array equals the
packaged one (so the folder is a faithful image of the .npz).  `oracle/gen_golden_citi_bike*.py` call it on demand.

    bash oracle/build_ref.sh /tmp/oracle && python oracle/setup_toy_topologies.py --maro /tmp/oracle/maro_src --home /tmp/oracle/home
    python oracle/setup_toy_topologies.py --check   # additionally regenerates every toy golden into a temp dir and diffs it
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

TOYS = ("toy.3s_4t", "toy.3s_tight", "toy.4s_4t", "toy.5s_6t", "toy.5s_filters", "toy.5s_win0")
# Monday 2019-01-07 00:00 America/New_York (EST, UTC-5) as UTC seconds; a 31-day span from any day of that week stays clear of the
# DST switch (2019-03-10), so tick -> local date is a plain division by 1440
MONDAY_EST = 1546837200


def start_utc_of(data) -> int:
    """The start instant that reproduces `tick_day` / `day_weekday`: the first local day has weekday day_weekday[0] and ends after
    the first `c` ticks (the toy generator starts at 00:00 UTC = 19:00 local: c = 300)."""
    td = data.tick_day
    change = np.flatnonzero(np.diff(td)) + 1
    c = int(change[0]) if len(change) else len(td)
    assert 0 < c <= 1440
    return MONDAY_EST + int(data.day_weekday[0]) * 86400 + (1440 - c) % 1440 * 60


def weather_rows(data, start_utc):
    """One row per local day at local noon: (utc timestamp, weather type, temperature).  The frame stores temperature as int16
    (truncation of the file's float), so the packaged integer is written as the float."""
    day0 = start_utc - ((start_utc - MONDAY_EST) % 86400)      # local midnight of the first day
    return [(day0 + d * 86400 + 43200, int(data.day_weather[d]), float(data.day_temperature[d])) for d in range(len(data.day_weather))]


def write_back(name, maro_root, home, quiet=False):
    import yaml

    from maro_amd.citi_bike.data import load_build_folder, load_topology
    from maro_amd.citi_bike.synthetic import write_build_folder
    data = load_topology(name)
    assert not data.day_holiday.any(), "the packaged toys were built with the holidays stub (no holidays)"
    bd = os.path.join(home, ".maro", "data", "citi_bike", ".build", name)
    t0 = start_utc_of(data)
    cfg = write_build_folder(data, bd, t0, weather=weather_rows(data, t0))
    back = load_build_folder(cfg, bd, name=name, is_holiday=None)   # the reference runs here with the `holidays` stand-in: no holidays
    for k in ("trip_tick", "trip_src", "trip_dst", "trip_duration", "capacity", "init_bikes", "station_id", "distance", "tick_day",
              "day_weekday", "day_holiday", "day_weather", "day_temperature"):
        a, b = getattr(data, k), getattr(back, k)
        assert a.shape == b.shape and np.array_equal(a, b), f"{name}: {k} does not survive the write-back"
    for k in ("resolution", "time_mean", "time_std", "supply_water_mark_ratio", "demand_water_mark_ratio", "scope_low_ratio",
              "scope_high_ratio", "extra_cost_mode", "filters"):
        assert getattr(data, k) == getattr(back, k), f"{name}: {k}"
    with open(os.path.join(bd, ".written_back_from_npz"), "wt") as fp:   # (ensure_toy: a folder the reference's own unseeded toy generator made is other data)
        fp.write(name + "\n")
    if maro_root:
        tdir = os.path.join(maro_root, "maro", "simulator", "scenarios", "citi_bike", "topologies", name)
        os.makedirs(tdir, exist_ok=True)
        with open(os.path.join(tdir, "config.yml"), "wt") as fp:
            yaml.safe_dump(cfg, fp)
    if not quiet:
        print(f"{name}: {data.n_stations} stations, {len(data.trip_tick)} trips, {len(data.tick_day)} ticks -> {bd}")
    return bd


def ensure_toy(maro_root, home, name):
    """Called by the golden generators: make sure `name`'s build folder and config.yml exist (idempotent)."""
    if name not in TOYS:
        return
    bd = os.path.join(home, ".maro", "data", "citi_bike", ".build", name)
    cfg = os.path.join(maro_root, "maro", "simulator", "scenarios", "citi_bike", "topologies", name, "config.yml")
    if not (os.path.exists(os.path.join(bd, "trips.bin")) and os.path.exists(os.path.join(bd, ".written_back_from_npz")) and os.path.exists(cfg)):
        write_back(name, maro_root, home, quiet=True)


def check_goldens(maro_root, stubs, home):
    """Regenerate every toy golden from the written-back folders (a FRESH home, so nothing of an older build is used) and compare
    with tests/golden byte for byte."""
    import subprocess
    import tempfile
    sys.path.insert(0, HERE)
    from gen_golden_citi_bike import CASES
    out = tempfile.mkdtemp(prefix="mrx_toy_golden_")
    bad = 0
    for case, (topo, _, _) in CASES.items():
        if topo not in TOYS:
            continue
        subprocess.check_call([sys.executable, os.path.join(HERE, "gen_golden_citi_bike.py"), "--maro", maro_root, "--stubs", stubs, "--out", out,
                               "--case", case, "--worker"], env=dict(os.environ, MARO_ORACLE_HOME=home))
        new, old = np.load(os.path.join(out, case + ".npz")), np.load(os.path.join(REPO, "tests", "golden", case + ".npz"))
        same = set(new.files) == set(old.files) and all(np.array_equal(new[k], old[k]) for k in old.files if k != "meta")
        print(f"{case}: {'identical' if same else 'DIFFERS'} ({len(old.files)} arrays)")
        bad += not same
    if bad:
        raise SystemExit(f"{bad} toy golden(s) differ")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src", help="the built reference checkout (oracle/build_ref.sh); '' = folders only")
    ap.add_argument("--home", default=os.environ.get("MARO_ORACLE_HOME", "/tmp/oracle/home"))
    ap.add_argument("--stubs", default="/tmp/oracle/stubs")
    ap.add_argument("--check", action="store_true", help="also regenerate the toy goldens from the written-back folders and diff them")
    a = ap.parse_args()
    if a.check:
        import tempfile
        a.home = tempfile.mkdtemp(prefix="mrx_toy_home_")     # nothing of an earlier build may leak into the comparison
        os.makedirs(os.path.join(a.home, ".maro"), exist_ok=True)
    for name in TOYS:
        write_back(name, a.maro, a.home)
    if a.check:
        check_goldens(a.maro, a.stubs, a.home)


if __name__ == "__main__":
    main()
