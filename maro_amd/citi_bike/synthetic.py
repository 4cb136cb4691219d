"""Synthetic citi_bike data sets (the counterpart of the reference's toy generator, ``maro/cli/data_pipeline/citi_bike.py``
``CitiBikeToyPipeline``: stations, a distance table and random trips written without any download).  The real ``ny.*``
topologies are built from downloads that are not available offline; ``city_data`` produces data of their SHAPE — a few
hundred stations with sparse neighbour lists, hub-and-rush-hour traffic, the default filter chain — for parity tests at that
size and for bench lines that are not about five-station toys.  ``python -m maro_amd.citi_bike.synthetic`` regenerates the
packaged ``city.*`` topologies (seeded)."""
import os

import numpy as np

from .data import _PKG_DIR, CitiBikeData


NY_FILTERS = [dict(type=0, num=80, windows=0), dict(type=1, num=40, windows=0), dict(type=2, num=20, windows=10)]   # topologies/ny.201801/config.yml:21-26


def city_data(rng, S=180, T=480, trips_per_tick=5.0, name=None, K=24, n_hubs=9, hub_weight=25.0, filters=None):
    """A city-shaped synthetic data set (the real NYC topologies are not shippable): a few hundred stations on a plane,
    distances = Euclidean to the K nearest stations only (the rest 0.0 = "not a neighbour", like distance_adj.csv rows with
    their 20 closest), trips drawn towards a handful of hubs with a rush-hour intensity, the reference's default filter
    chain (distance 20 -> requirements 10 -> trip_window 6, windows 10).  S > 64 puts the engines on their generic
    (non register-frame) station loops and several decision-mask words."""
    day_len = min(T, 1440) if T % 1440 == 0 else 300
    n_days = (T + day_len - 1) // day_len
    xy = rng.uniform(0, 10, (S, 2))
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    dist = np.zeros((S, S))
    for s in range(S):
        nb = np.argsort(d[s], kind="stable")[1:K + 1]
        dist[s, nb] = np.round(d[s, nb], 4) + 1e-4
    hubs = rng.choice(S, n_hubs, replace=False)
    w = np.ones(S)
    w[hubs] = hub_weight
    w /= w.sum()
    tod = np.arange(T) % day_len
    lam = trips_per_tick * (0.4 + 1.2 * np.exp(-((tod - day_len * 0.45) / (day_len * 0.18)) ** 2))
    per = rng.poisson(lam)
    tick = np.repeat(np.arange(T), per).astype(np.int32)
    n = len(tick)
    src = rng.choice(S, n, p=w).astype(np.int32)
    dst = rng.choice(S, n, p=w[::-1] / w[::-1].sum()).astype(np.int32)
    cap = rng.randint(8, 60, S).astype(np.int32)
    init = (cap * rng.uniform(0.1, 0.9, S)).astype(np.int32)
    return CitiBikeData(
        name=name or "city.%ds" % S, trip_tick=tick, trip_src=src, trip_dst=dst, trip_duration=rng.randint(1, 45, n).astype(np.int32),
        capacity=cap, init_bikes=init, station_id=np.arange(S, dtype=np.int32) + 3000, distance=dist,
        tick_day=(np.arange(T) // day_len).astype(np.int32), day_weekday=(np.arange(n_days) % 7).astype(np.int16), day_holiday=np.zeros(n_days, np.int16),
        day_weather=(np.arange(n_days) * 3 % 4).astype(np.int16), day_temperature=(21 - 9 * (np.arange(n_days) % 3)).astype(np.int16), resolution=20, time_mean=20.0, time_std=5.0,
        supply_water_mark_ratio=0.8, demand_water_mark_ratio=0.2, scope_low_ratio=0.0, scope_high_ratio=1.0, extra_cost_mode=0,
        filters=filters or [dict(type=0, num=20, windows=0), dict(type=1, num=10, windows=0), dict(type=2, num=6, windows=10)])


FILTER_NAME = {0: "distance", 1: "requirements", 2: "trip_window"}


def write_build_folder(data: CitiBikeData, build_dir: str, start_utc: int, weather=None, rng=None) -> dict:
    """Write `data` as the folder ``maro data build`` would leave behind (trips.bin + KNYC_daily.bin in MARO's binary format
    via ``maro_amd.data_lib.write_binary``, station_meta.csv, distance_adj.csv) and return the matching ``config.yml``
    dict: the reference's own ``Env("citi_bike", ...)`` can then run the synthetic topology, and
    ``data.load_build_folder`` reads it back into the packaged form (calendar features from the real dates).  Tick t of
    `data` becomes the minute starting at ``start_utc + 60 t`` (`rng` scatters the seconds inside the minute; file order =
    trip order).  `weather` = rows (utc timestamp, weather type, temperature); default one row per day."""
    import csv

    from ..data_lib import write_binary
    os.makedirs(build_dir, exist_ok=True)
    n = len(data.trip_tick)
    sec = np.zeros(n, np.int64) if rng is None else np.sort(rng.randint(0, 60, n) + data.trip_tick.astype(np.int64) * 60) - data.trip_tick.astype(np.int64) * 60
    ts = start_utc + data.trip_tick.astype(np.int64) * 60 + sec
    T = len(data.tick_day)
    write_binary(os.path.join(build_dir, "trips.bin"),
                 dict(timestamp=ts, durations=data.trip_duration, src_station=data.trip_src, dest_station=data.trip_dst),
                 dict(timestamp="i8", durations="i", src_station="i", dest_station="i"), starttime=start_utc, endtime=start_utc + (T - 2) * 60,
                 raw_names=dict(timestamp="start_time", durations="duration", src_station="start_station_index", dest_station="end_station_index"))
    n_days = T // 1440 + 2
    if weather is None:
        weather = [(start_utc - start_utc % 86400 + d * 86400 + 43200, (d * 3) % 4, 21.5 - 9.25 * (d % 3)) for d in range(n_days)]
    w = np.array(weather, np.float64).reshape(-1, 3)
    write_binary(os.path.join(build_dir, "KNYC_daily.bin"), dict(timestamp=w[:, 0].astype(np.int64), weather=w[:, 1].astype(np.int32), temp=w[:, 2].astype(np.float32)),
                 dict(timestamp="i8", weather="i", temp="f"), raw_names=dict(timestamp="date", weather="weather", temp="temp"))
    with open(os.path.join(build_dir, "station_meta.csv"), "wt", newline="") as fp:
        wr = csv.writer(fp)
        wr.writerow(["station_index", "capacity", "init", "latitude", "longitude", "station_id"])
        for s in range(data.n_stations):
            wr.writerow([s, int(data.capacity[s]), int(data.init_bikes[s]), 40.7 + 0.001 * s, -74.0 + 0.001 * s, int(data.station_id[s])])
    with open(os.path.join(build_dir, "distance_adj.csv"), "wt", newline="") as fp:
        wr = csv.writer(fp)
        wr.writerow(list(range(data.n_stations)))
        for row in data.distance:
            wr.writerow([repr(float(x)) for x in row])
    return dict(
        decision=dict(extra_cost_mode=["source", "target"][data.extra_cost_mode], resolution=data.resolution,
                      effective_time_mean=data.time_mean, effective_time_std=data.time_std,
                      supply_water_mark_ratio=data.supply_water_mark_ratio, demand_water_mark_ratio=data.demand_water_mark_ratio,
                      action_scope=dict(low=data.scope_low_ratio, high=data.scope_high_ratio,
                                        filters=[dict(type=FILTER_NAME[f["type"]], num=f["num"], **({"windows": f["windows"]} if f["type"] == 2 else {}))
                                                 for f in data.filters])),
        time_zone="America/New_York",
        **{k: os.path.join(build_dir, f) for k, f in (("trip_data", "trips.bin"), ("weather_data", "KNYC_daily.bin"),
                                                      ("stations_init_data", "station_meta.csv"), ("distance_adj_data", "distance_adj.csv"))})


# name -> generator arguments; start_utc 2019-06-03 04:00 UTC = Monday 00:00 in New York (EDT)
PACKAGED = {"city.180s": dict(seed=20260924, S=180, T=1440 * 2, trips_per_tick=6.0, start_utc=1559534400),
            # the SIZE of the reference's shipped ny.* topologies (topologies/ny.201801/config.yml): ~800 stations, a month of
            # 1-minute ticks (44 640), ~1.5 M trips, the ny filter chain distance 80 -> requirements 40 -> trip_window 20 over 10
            # windows, decision resolution 20.  6.8 MB as .npz: generated (seeded, ~2 s) by __graft_entry__.build() / on first
            # use instead of being committed.
            "city.800s": dict(seed=20260925, S=800, T=1440 * 31, trips_per_tick=43.0, start_utc=1559534400, K=100, n_hubs=40, hub_weight=8.0,
                              filters=NY_FILTERS)}
GENERATED_ON_DEMAND = ("city.800s",)


def build_packaged(name: str, build_dir: str):
    """(config dict, CitiBikeData): the packaged topology `name`, generated, written as a build folder and read back."""
    from .data import load_build_folder
    kw = dict(PACKAGED[name])
    rng = np.random.RandomState(kw.pop("seed"))
    start_utc = kw.pop("start_utc")
    raw = city_data(rng, name=name, **kw)
    cfg = write_build_folder(raw, build_dir, start_utc, rng=rng)
    return cfg, load_build_folder(cfg, build_dir, name=name, is_holiday=None)   # (the goldens of these sets came from the reference with its no-holidays stand-in)


def ensure_packaged(name: str) -> str:
    """Path of the packaged .npz of `name`, generating it first when it is one of the on-demand topologies."""
    path = os.path.join(_PKG_DIR, name + ".npz")
    if not os.path.exists(path) and name in GENERATED_ON_DEMAND:
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp(prefix="mrx_city_")
        try:
            _, data = build_packaged(name, tmp)
            part = path + f".{os.getpid()}.tmp.npz"
            data.save(part)
            os.replace(part, path)   # atomic: several ranks / test workers may get here together
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return path


def main():
    import argparse
    import tempfile

    import yaml
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-root", help="keep the build folders (and config.yml) here, e.g. for the reference's Env; default: a temp dir")
    a = ap.parse_args()
    root = a.build_root or tempfile.mkdtemp(prefix="mrx_city_")
    for name in PACKAGED:
        bd = os.path.join(root, name)
        cfg, data = build_packaged(name, bd)
        with open(os.path.join(bd, "config.yml"), "wt") as fp:
            yaml.safe_dump(cfg, fp)
        path = os.path.join(_PKG_DIR, name + ".npz")
        data.save(path)
        print(f"{name}: {data.n_stations} stations, {len(data.trip_tick)} trips, {len(data.day_weekday)} days -> {path} ({os.path.getsize(path) // 1024} KiB); build folder {bd}")


if __name__ == "__main__":
    main()
