"""Compiled citi_bike scenario data: everything the reference reads from a topology folder
(``maro/simulator/scenarios/citi_bike/business_engine.py:205-260``: config.yml, trips.bin via BinaryReader,
KNYC_daily.bin via WeatherTable, station_meta.csv, distance_adj.csv) flattened into arrays.

The flat form is produced by ``load_build_folder`` below — a native reader of the topology's ``config.yml`` and of the
built data folder (``maro_amd.data_lib`` reads MARO's binary format; no MARO checkout needed) — and shipped as
``topologies/<name>.npz``; the trip table is shared by every env of a batch and lives in HBM as SoA columns.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "topologies")

FILTER_DISTANCE, FILTER_REQUIREMENTS, FILTER_TRIP_WINDOW = 0, 1, 2
EXTRA_COST_SOURCE, EXTRA_COST_TARGET, EXTRA_COST_NEIGHBORS = 0, 1, 2


@dataclass
class CitiBikeData:
    name: str
    # trips, sorted by tick then file order (ItemTickPicker, data_lib/binary_reader.py:80-112; 1 tick = 1 minute)
    trip_tick: np.ndarray      # int32 [n]
    trip_src: np.ndarray       # int32 [n]
    trip_dst: np.ndarray       # int32 [n]
    trip_duration: np.ndarray  # int32 [n]  (ticks)
    # stations (stations_info.py:19-37)
    capacity: np.ndarray       # int32 [S]
    init_bikes: np.ndarray     # int32 [S]
    station_id: np.ndarray     # int32 [S]
    distance: np.ndarray       # float64 [S, S]  (adj_loader.py; 0.0 = not a neighbour)
    # calendar features per day, already cast the way the int16 frame attributes store them
    tick_day: np.ndarray       # int32 [max_tick_available] day index of each tick (business_engine.py:367-369)
    day_weekday: np.ndarray    # int16 [n_days]
    day_holiday: np.ndarray    # int16 [n_days]
    day_weather: np.ndarray    # int16 [n_days]
    day_temperature: np.ndarray  # int16 [n_days]
    # decision strategy (decision_strategy.py:181-211)
    resolution: int = 20
    time_mean: float = 20.0
    time_std: float = 5.0
    supply_water_mark_ratio: float = 0.8
    demand_water_mark_ratio: float = 0.2
    scope_low_ratio: float = 0.0
    scope_high_ratio: float = 1.0
    extra_cost_mode: int = EXTRA_COST_SOURCE
    filters: List[dict] = field(default_factory=list)   # [{"type": 0|1|2, "num": int, "windows": int}]

    @property
    def n_stations(self) -> int:
        return int(self.capacity.shape[0])

    def trip_offsets(self, max_tick: int) -> np.ndarray:
        """CSR offsets: trips of tick t are rows [off[t], off[t+1])."""
        off = np.searchsorted(self.trip_tick, np.arange(max_tick + 1), side="left").astype(np.int32)
        return off

    def neighbors(self) -> np.ndarray:
        """int32 [S, S]: neighbours of each station sorted by distance (dist != 0), -1 padded
        (decision_strategy.py:381-391: `sorted(..., key=dist)` is stable, ties keep index order)."""
        S = self.n_stations
        out = -np.ones((S, S), np.int32)
        for s in range(S):
            idx = np.flatnonzero(self.distance[s] != 0.0)
            out[s, :len(idx)] = idx[np.argsort(self.distance[s][idx], kind="stable")]   # stable: ties keep index order
        return out

    # ---- packaged form
    def save(self, path: str):
        meta = dict(name=self.name, resolution=self.resolution, time_mean=self.time_mean, time_std=self.time_std,
                    supply_water_mark_ratio=self.supply_water_mark_ratio, demand_water_mark_ratio=self.demand_water_mark_ratio,
                    scope_low_ratio=self.scope_low_ratio, scope_high_ratio=self.scope_high_ratio,
                    extra_cost_mode=self.extra_cost_mode, filters=self.filters)
        arrays = {k: getattr(self, k) for k in ("trip_tick", "trip_src", "trip_dst", "trip_duration", "capacity", "init_bikes",
                                                "station_id", "distance", "tick_day", "day_weekday", "day_holiday",
                                                "day_weather", "day_temperature")}
        np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), np.uint8), **arrays)

    @staticmethod
    def load(path: str) -> "CitiBikeData":
        z = np.load(path)
        meta = json.loads(bytes(z["meta"]).decode())
        return CitiBikeData(**meta, **{k: z[k] for k in z.files if k != "meta"})


def load_build_folder(config, build_dir: str, name: str = None, is_holiday="us") -> CitiBikeData:
    """Compile a citi_bike topology natively: `config` = the topology's config.yml (path or dict: ``decision`` block,
    ``time_zone``), `build_dir` = the folder ``maro data build`` / the toy generator wrote (trips.bin, KNYC_daily.bin,
    station_meta.csv, distance_adj.csv).  Mirrors what ``CitibikeBusinessEngine`` reads at construction
    (``business_engine.py:205-260``): trips through ``ItemTickPicker`` with one tick per minute
    (``data_lib/binary_reader.py:80-112`` -> ``maro_amd.data_lib.pick_ticks``), stations (``stations_info.py:19-37``), the
    distance matrix (``adj_loader.py``, first row skipped), the weather table keyed by local date (``weather_table.py:29-41``),
    and per tick the local date's weekday / holiday flag (``business_engine.py:367-392``).  `is_holiday`: the reference asks
    ``holidays.US()``; the default "us" is that calendar restated natively (``us_calendar.is_us_holiday``: the federal holidays
    and their observed days).  A callable ``(date) -> bool`` replaces it; None knows no holidays — the mode the packaged toy
    topologies and every citi_bike golden were generated in (the reference ran there with a `holidays` stand-in that contains
    nothing: oracle/build_ref.sh), so regenerating or re-checking those passes None."""
    import csv
    from datetime import datetime, timedelta, timezone

    import yaml
    from dateutil.tz import gettz

    from ..data_lib import pick_ticks, read_binary
    if isinstance(is_holiday, str):
        if is_holiday != "us":
            raise ValueError(f"is_holiday: unknown calendar {is_holiday!r} (\"us\", a callable, or None)")
        from .us_calendar import is_us_holiday as is_holiday
    if not isinstance(config, dict):
        with open(config, "rt") as fp:
            config = yaml.safe_load(fp)
    tz = gettz(config["time_zone"])
    hdr, rec = read_binary(os.path.join(build_dir, "trips.bin"))
    t0 = int(hdr["starttime"])
    n_ticks = int((int(hdr["endtime"]) - t0) // 60) + 2
    tick = pick_ticks(rec["timestamp"], t0, n_ticks, "m")
    keep = tick >= 0
    trips = np.stack([tick[keep], rec["src_station"][keep], rec["dest_station"][keep], rec["durations"][keep]], axis=1).astype(np.int32)
    with open(os.path.join(build_dir, "station_meta.csv"), "rt") as fp:
        st = sorted(((int(r["station_index"]), int(r["init"]), int(r["capacity"]), int(float(r["station_id"]))) for r in csv.DictReader(fp)))
    with open(os.path.join(build_dir, "distance_adj.csv"), "rt") as fp:
        rows = list(csv.reader(fp))[1:]
    dist = np.array([[float(c) for c in row] for row in rows], np.float64).reshape(len(st), len(st))
    _, wrec = read_binary(os.path.join(build_dir, "KNYC_daily.bin"))
    weather = {}
    for ts, w, temp in zip(wrec["timestamp"].tolist(), wrec["weather"].tolist(), wrec["temp"].tolist()):
        weather[datetime.fromtimestamp(ts, timezone.utc).astimezone(tz).date()] = (w, temp)   # later rows win, like the dict in the reference
    start = datetime.fromtimestamp(t0, timezone.utc).astimezone(tz)
    days, tick_day, feats = {}, [], []
    for t in range(n_ticks):
        d = (start + timedelta(minutes=t)).date()          # relativedelta(minutes=t) on an aware datetime: wall-clock arithmetic
        if d not in days:
            days[d] = len(days)
            w = weather.get(d)
            feats.append((d.weekday(), int(bool(is_holiday(d))) if is_holiday else 0, 0 if w is None else w[0], 0 if w is None else w[1]))
        tick_day.append(days[d])
    dec = config["decision"]
    ftype = {"distance": FILTER_DISTANCE, "requirements": FILTER_REQUIREMENTS, "trip_window": FILTER_TRIP_WINDOW}
    return CitiBikeData(
        name=name or os.path.basename(os.path.normpath(build_dir)), trip_tick=trips[:, 0].copy(), trip_src=trips[:, 1].copy(),
        trip_dst=trips[:, 2].copy(), trip_duration=trips[:, 3].copy(),
        capacity=np.array([x[2] for x in st], np.int32), init_bikes=np.array([x[1] for x in st], np.int32),
        station_id=np.array([x[3] for x in st], np.int32), distance=dist, tick_day=np.array(tick_day, np.int32),
        day_weekday=np.array([f[0] for f in feats]).astype(np.int16), day_holiday=np.array([f[1] for f in feats]).astype(np.int16),
        day_weather=np.array([f[2] for f in feats]).astype(np.int16),
        day_temperature=np.array([f[3] for f in feats], np.float64).astype(np.int16),   # float -> i2 attribute: numpy truncation
        resolution=int(dec["resolution"]), time_mean=float(dec["effective_time_mean"]), time_std=float(dec["effective_time_std"]),
        supply_water_mark_ratio=float(dec["supply_water_mark_ratio"]), demand_water_mark_ratio=float(dec["demand_water_mark_ratio"]),
        scope_low_ratio=float(dec["action_scope"]["low"]), scope_high_ratio=float(dec["action_scope"]["high"]),
        extra_cost_mode={"source": EXTRA_COST_SOURCE, "target": EXTRA_COST_TARGET}[dec["extra_cost_mode"]],  # common.py:155-160
        filters=[dict(type=ftype[f["type"]], num=int(f["num"]), windows=int(f.get("windows", 0))) for f in dec["action_scope"]["filters"]])


def available_topologies() -> List[str]:
    return sorted(f[:-4] for f in os.listdir(_PKG_DIR) if f.endswith(".npz")) if os.path.isdir(_PKG_DIR) else []


def load_topology(name_or_path: str) -> CitiBikeData:
    if os.path.isfile(name_or_path):
        return CitiBikeData.load(name_or_path)
    p = os.path.join(_PKG_DIR, name_or_path + ".npz")
    if not os.path.exists(p):
        from .synthetic import GENERATED_ON_DEMAND, ensure_packaged
        if name_or_path in GENERATED_ON_DEMAND:
            ensure_packaged(name_or_path)   # seeded, ~2 s (too large to commit)
    if os.path.exists(p):
        return CitiBikeData.load(p)
    raise FileNotFoundError(f"unknown citi_bike topology {name_or_path!r}; packaged: {available_topologies()}")
