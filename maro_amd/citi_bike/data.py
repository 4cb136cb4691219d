"""Compiled citi_bike scenario data: everything the reference reads from a topology folder
(``maro/simulator/scenarios/citi_bike/business_engine.py:205-260``: config.yml, trips.bin via BinaryReader,
KNYC_daily.bin via WeatherTable, station_meta.csv, distance_adj.csv) flattened into arrays.

The flat form is produced once by ``tools/import_maro_citi_bike.py`` (which needs a MARO checkout: the on-disk
binary format is read with the reference's own reader) and shipped as ``topologies/<name>.npz``; the trip table
is shared by every env of a batch and lives in HBM as SoA columns.
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
            nb = [(i, d) for i, d in enumerate(self.distance[s]) if d != 0.0]
            nb.sort(key=lambda kv: kv[1])
            out[s, :len(nb)] = [i for i, _ in nb]
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


def available_topologies() -> List[str]:
    return sorted(f[:-4] for f in os.listdir(_PKG_DIR) if f.endswith(".npz")) if os.path.isdir(_PKG_DIR) else []


def load_topology(name_or_path: str) -> CitiBikeData:
    if os.path.isfile(name_or_path):
        return CitiBikeData.load(name_or_path)
    p = os.path.join(_PKG_DIR, name_or_path + ".npz")
    if os.path.exists(p):
        return CitiBikeData.load(p)
    raise FileNotFoundError(f"unknown citi_bike topology {name_or_path!r}; packaged: {available_topologies()}")
