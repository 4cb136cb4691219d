#!/usr/bin/env python3
"""Compile a built MARO citi_bike topology into the engine's flat .npz form.

    python tools/import_maro_citi_bike.py --maro /tmp/oracle/maro_src --topology toy.3s_4t \
        --build ~/.maro/data/citi_bike/.build/toy.3s_4t [--stubs /tmp/oracle/stubs]

Reads the topology's config.yml from the MARO checkout and trips.bin / KNYC_daily.bin / station_meta.csv /
distance_adj.csv from the build folder, using the reference's own BinaryReader / ItemTickPicker / WeatherTable
(the binary on-disk format is not re-implemented here).  `holidays` comes from whatever module is importable
(a stub that knows no holidays where the real package is absent — then day_holiday is all zero, matching the
goldens generated in the same environment).
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", required=True)
    ap.add_argument("--topology", default="toy.3s_4t")
    ap.add_argument("--build", required=True)
    ap.add_argument("--stubs")
    ap.add_argument("--out", default=os.path.join(REPO, "maro_amd", "citi_bike", "topologies"))
    args = ap.parse_args()
    if args.stubs:
        sys.path.insert(0, args.stubs)
    sys.path.insert(0, args.maro)
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    import holidays
    import yaml
    from dateutil.relativedelta import relativedelta
    from dateutil.tz import gettz
    from maro.data_lib import BinaryReader
    from maro.simulator.scenarios.citi_bike.adj_loader import load_adj_from_csv
    from maro.simulator.scenarios.citi_bike.stations_info import get_station_info
    from maro.simulator.scenarios.citi_bike.weather_table import WeatherTable

    from maro_amd.citi_bike.data import CitiBikeData

    with open(os.path.join(args.maro, "maro/simulator/scenarios/citi_bike/topologies", args.topology, "config.yml")) as fp:
        conf = yaml.safe_load(fp)
    build = os.path.expanduser(args.build)
    tz = gettz(conf["time_zone"])
    reader = BinaryReader(os.path.join(build, "trips.bin"))
    start = reader.start_datetime.astimezone(tz)
    n_ticks = int((reader.header.endtime - reader.header.starttime) // 60) + 2
    picker = reader.items_tick_picker(0, n_ticks, time_unit="m")
    rows = []
    for tick in range(n_ticks):
        for it in picker.items(tick):
            rows.append((tick, it.src_station, it.dest_station, it.durations))
    trips = np.array(rows, np.int32).reshape(-1, 4)
    st = sorted(get_station_info(os.path.join(build, "station_meta.csv")), key=lambda s: s.index)
    dist = np.array(load_adj_from_csv(os.path.join(build, "distance_adj.csv"), skiprows=1), np.float64)
    weather = WeatherTable(os.path.join(build, "KNYC_daily.bin"), tz)
    us_holidays = holidays.US()
    days, tick_day, feats = {}, [], []
    for tick in range(n_ticks):
        d = (start + relativedelta(minutes=tick)).date()          # business_engine.py:367-369
        if d not in days:
            days[d] = len(days)
            w = weather[d]
            feats.append((d.weekday(), int(d in us_holidays), 0 if w is None else w.weather, 0 if w is None else w.temp))
        tick_day.append(days[d])
    dec = conf["decision"]
    ftype = {"distance": 0, "requirements": 1, "trip_window": 2}
    data = CitiBikeData(
        name=args.topology, trip_tick=trips[:, 0].copy(), trip_src=trips[:, 1].copy(), trip_dst=trips[:, 2].copy(),
        trip_duration=trips[:, 3].copy(),
        capacity=np.array([s.capacity for s in st], np.int32), init_bikes=np.array([s.bikes for s in st], np.int32),
        station_id=np.array([s.id for s in st], np.int32), distance=dist.reshape(len(st), len(st)),
        tick_day=np.array(tick_day, np.int32),
        day_weekday=np.array([f[0] for f in feats]).astype(np.int16), day_holiday=np.array([f[1] for f in feats]).astype(np.int16),
        day_weather=np.array([f[2] for f in feats]).astype(np.int16),
        day_temperature=np.array([f[3] for f in feats], np.float64).astype(np.int16),   # float -> i2 attribute: numpy truncation
        resolution=int(dec["resolution"]), time_mean=float(dec["effective_time_mean"]), time_std=float(dec["effective_time_std"]),
        supply_water_mark_ratio=float(dec["supply_water_mark_ratio"]), demand_water_mark_ratio=float(dec["demand_water_mark_ratio"]),
        scope_low_ratio=float(dec["action_scope"]["low"]), scope_high_ratio=float(dec["action_scope"]["high"]),
        extra_cost_mode={"source": 0, "target": 1}[dec["extra_cost_mode"]],  # common.py:155-160 (target_neighbors is commented out there)
        filters=[dict(type=ftype[f["type"]], num=int(f["num"]), windows=int(f.get("windows", 0))) for f in dec["action_scope"]["filters"]])
    os.makedirs(args.out, exist_ok=True)
    path = os.path.join(args.out, args.topology + ".npz")
    data.save(path)
    print(f"{args.topology}: {len(trips)} trips, {len(st)} stations, {len(feats)} days -> {path} ({os.path.getsize(path) // 1024} KiB)")


if __name__ == "__main__":
    main()
