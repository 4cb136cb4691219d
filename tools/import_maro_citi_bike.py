#!/usr/bin/env python3
"""Compile a built MARO citi_bike topology into the engine's flat .npz form (maro_amd/citi_bike/data.py::load_build_folder).
Native: config.yml, trips.bin / KNYC_daily.bin (MARO's binary format, read by maro_amd.data_lib), station_meta.csv and
distance_adj.csv are parsed by maro_amd itself, no MARO checkout needed.

    python tools/import_maro_citi_bike.py --config <topology>/config.yml --build ~/.maro/data/citi_bike/.build/toy.3s_4t --name toy.3s_4t

day_holiday: the `holidays` package when it is importable, else the native restatement of holidays.US()
(maro_amd/citi_bike/us_calendar.py); --no-holidays: all zero, as in the packaged toy topologies and their goldens (the reference
generated those with a stand-in that contains nothing).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--build", required=True)
    ap.add_argument("--name")
    ap.add_argument("--out", default=os.path.join(REPO, "maro_amd", "citi_bike", "topologies"))
    ap.add_argument("--no-holidays", action="store_true")
    args = ap.parse_args()
    from maro_amd.citi_bike.data import load_build_folder
    if args.no_holidays:
        is_holiday = None
    else:
        try:
            import holidays
            us = holidays.US()
            is_holiday = (lambda d: d in us) if len(holidays.US(years=2019)) else "us"  # noqa: E731  (an empty stand-in: use the native calendar)
        except ImportError:
            is_holiday = "us"
    name = args.name or os.path.basename(os.path.normpath(args.build))
    data = load_build_folder(args.config, os.path.expanduser(args.build), name=name, is_holiday=is_holiday)
    os.makedirs(args.out, exist_ok=True)
    path = os.path.join(args.out, name + ".npz")
    data.save(path)
    print(f"{name}: {len(data.trip_tick)} trips, {data.n_stations} stations, {len(data.day_weekday)} days -> {path} ({os.path.getsize(path) // 1024} KiB)")


if __name__ == "__main__":
    main()
