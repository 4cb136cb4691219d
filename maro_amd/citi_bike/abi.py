"""ctypes mirrors of include/maro_amd_citi_bike.h (shared by the product engine and the tests' CPU harness)."""
from __future__ import annotations

import ctypes

import numpy as np

from .data import CitiBikeData

MAX_FILTERS = 4
STATION_ATTRS = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "id", "weekday", "temperature",
                 "weather", "holiday", "extra_cost", "transfer_cost", "failed_return", "min_bikes"]
MATRIX_ATTRS = ["trips_adj"]
NODE_ATTRS = {"stations": STATION_ATTRS, "matrices": MATRIX_ATTRS}
NODE_TYPE = {"stations": 0, "matrices": 1}
HDR_TICK, HDR_FLAGS, HDR_STATUS, HDR_WORDS = 0, 1, 13, 16

_i32p, _f64p, _i16p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int16)


class MrxCbTopology(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_stations", "n_trips", "n_ticks", "n_days")]
                + [(n, _i32p) for n in ("trip_tick", "trip_src", "trip_dst", "trip_duration", "capacity", "init_bikes",
                                        "station_id")]
                + [("distance", _f64p), ("tick_day", _i32p)]
                + [(n, _i16p) for n in ("day_weekday", "day_holiday", "day_weather", "day_temperature")]
                + [("resolution", ctypes.c_int32)]
                + [(n, ctypes.c_double) for n in ("supply_water_mark_ratio", "demand_water_mark_ratio", "scope_low_ratio",
                                                  "scope_high_ratio")]
                + [("extra_cost_mode", ctypes.c_int32), ("n_filters", ctypes.c_int32)]
                + [(n, ctypes.c_int32 * MAX_FILTERS) for n in ("filter_type", "filter_num", "filter_windows")])


class MrxCbConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_envs", "device", "start_tick", "durations", "snapshot_resolution",
                                              "max_snapshots", "max_actions", "delivery_capacity", "transfer_times_cap", "decision_mode")]


class MrxCbLayout(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_envs", "env_stride", "n_stations", "frame_words", "ring_slots", "scope_cap",
                                               "delivery_capacity", "transfer_times_cap", "env_major", "reserved0")]
                + [(n, ctypes.c_int64) for n in ("off_hdr", "off_live", "off_ring", "off_ring_fi", "off_transfer_times",
                                                 "workspace_bytes", "off_prof")])


def topology_struct(d: CitiBikeData):
    """(struct, keepalive): flat mrx_cb_topology over the arrays of a CitiBikeData."""
    keep = {}

    def arr(name, dtype, ptr):
        a = np.ascontiguousarray(getattr(d, name), dtype)
        keep[name] = a
        return a.ctypes.data_as(ptr)

    t = MrxCbTopology()
    t.n_stations, t.n_trips, t.n_ticks, t.n_days = d.n_stations, len(d.trip_tick), len(d.tick_day), len(d.day_weekday)
    for n in ("trip_tick", "trip_src", "trip_dst", "trip_duration", "capacity", "init_bikes", "station_id", "tick_day"):
        setattr(t, n, arr(n, np.int32, _i32p))
    t.distance = arr("distance", np.float64, _f64p)
    for n in ("day_weekday", "day_holiday", "day_weather", "day_temperature"):
        setattr(t, n, arr(n, np.int16, _i16p))
    t.resolution = int(d.resolution)
    t.supply_water_mark_ratio, t.demand_water_mark_ratio = float(d.supply_water_mark_ratio), float(d.demand_water_mark_ratio)
    t.scope_low_ratio, t.scope_high_ratio = float(d.scope_low_ratio), float(d.scope_high_ratio)
    t.extra_cost_mode = int(d.extra_cost_mode)
    if len(d.filters) > MAX_FILTERS:
        raise ValueError(f"at most {MAX_FILTERS} neighbour filters are supported")
    t.n_filters = len(d.filters)
    for i, f in enumerate(d.filters):
        t.filter_type[i], t.filter_num[i], t.filter_windows[i] = int(f["type"]), int(f["num"]), int(f.get("windows", 0))
    return t, keep


def draw_transfer_times(d: CitiBikeData, seeds, count: int) -> np.ndarray:
    """int32 [len(seeds), count]: what `np.random.seed(seed)` followed by `count` reads of
    BikeDecisionStrategy.transfer_time yields (decision_strategy.py:213-216) — the reference draws them from the
    process-global numpy RNG, so a batch needs one pre-drawn stream per env."""
    out = np.empty((len(seeds), count), np.int32)
    for i, s in enumerate(seeds):
        rs = np.random.RandomState(int(s))
        # round() of a Python float is round-half-to-even, as is np.rint
        out[i] = np.rint(rs.normal(d.time_mean, d.time_std, size=count)).astype(np.int32)
    return out
