"""Randomised differential test for citi_bike (tooling + a pytest slice): synthetic data sets with 2..40 stations (more
than 32 exercises the second decision-mask word), random neighbour graphs, filters that cut, scope ratios, water marks and
transfer-time distributions; the host-compiled device code vs the pure-Python oracle.
`python tests/fuzz_citi_bike.py N [seed0]`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_data(rng):
    from maro_amd.citi_bike.data import CitiBikeData
    S = int(rng.choice([2, 3, 5, 8, 13, 33, 40]))
    T = 400
    n = int(rng.randint(200, 1500))
    tick = np.sort(rng.randint(0, T, n)).astype(np.int32)
    cap = rng.randint(2, 30, S).astype(np.int32)
    init = (cap * rng.uniform(0, 1, S)).astype(np.int32)
    dist = rng.uniform(0.1, 9.0, (S, S))
    dist = np.round((dist + dist.T) / 2, 3)
    dist[rng.rand(S, S) < 0.25] = 0.0            # not a neighbour (asymmetric on purpose)
    np.fill_diagonal(dist, 0.0)
    kinds = [[0], [0, 1], [0, 1, 2], [1], [2], [2, 1], [0, 2]][int(rng.randint(0, 7))]
    filters = [dict(type=k, num=int(rng.randint(1, 7)), windows=int(rng.choice([0, 1, 3, 10]))) for k in kinds]
    days = 2
    return CitiBikeData(
        name="fuzz", trip_tick=tick, trip_src=rng.randint(0, S, n).astype(np.int32), trip_dst=rng.randint(0, S, n).astype(np.int32),
        trip_duration=rng.randint(0, 50, n).astype(np.int32), capacity=cap, init_bikes=init, station_id=np.arange(S, dtype=np.int32) + 100,
        distance=dist, tick_day=(np.arange(T) // 250).astype(np.int32), day_weekday=np.array([2, 3], np.int16), day_holiday=np.array([0, 1], np.int16),
        day_weather=np.array([1, 0], np.int16), day_temperature=np.array([17, -3], np.int16), resolution=int(rng.choice([5, 20])),
        time_mean=float(rng.choice([2, 8, 20])), time_std=float(rng.choice([1, 3, 6])), supply_water_mark_ratio=float(rng.choice([0.6, 0.8])),
        demand_water_mark_ratio=float(rng.choice([0.2, 0.4])), scope_low_ratio=float(rng.choice([0, 0.15, 0.5])),
        scope_high_ratio=float(rng.choice([1, 0.85, 0.5])), extra_cost_mode=int(rng.randint(0, 2)), filters=filters)


def run_city_case(case_seed, S, backend=None, n_envs=3, **kw):
    from tests.cb_batch_check import run_batch_vs_oracle
    from tests.emu.cb_emu import CbEmuBackend
    B = backend or CbEmuBackend
    from maro_amd.citi_bike.synthetic import city_data
    data = city_data(np.random.RandomState(case_seed), S=S)
    kw = dict(dict(durations=400, snapshot_resolution=10), **kw)
    b = B(data, n_envs=n_envs, max_actions=1, **kw)
    return run_batch_vs_oracle(b, data, kw, seeds=np.arange(n_envs) + case_seed, episodes=1,
                               check_envs=sorted({0, n_envs // 2, n_envs - 1}))


def run_case(case_seed, backend=None):
    from tests.cb_batch_check import run_batch_vs_oracle
    from tests.emu.cb_emu import CbEmuBackend
    CbEmuBackend = backend or CbEmuBackend
    rng = np.random.RandomState(case_seed)
    data = random_data(rng)
    kw = dict(durations=int(rng.choice([150, 400])), snapshot_resolution=int(rng.choice([1, 4, 10])))
    if rng.rand() < 0.4:
        kw["max_snapshots"] = int(rng.randint(2, 12))
    if rng.rand() < 0.35 and kw["durations"] == 150:   # start ticks that are not multiples of the snapshot / decision resolutions
        kw["start_tick"] = int(rng.choice([7, 33, 101, 240]))
    b = CbEmuBackend(data, n_envs=3, max_actions=1, **kw)
    try:
        return run_batch_vs_oracle(b, data, kw, seeds=np.arange(3) + case_seed, episodes=1)
    except Exception:
        print("FAILING citi_bike fuzz case", case_seed, data.n_stations, data.filters, kw)
        raise


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for s in range(s0, s0 + n):
        run_case(s)
        if (s - s0 + 1) % 10 == 0:
            print(s - s0 + 1, "cases ok", flush=True)
    print("all", n, "cases ok")
