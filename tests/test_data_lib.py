"""Native readers of MARO's on-disk data (maro_amd/data_lib.py, CimTopology load_data_folder, citi_bike load_build_folder):
the product needs no MARO checkout to compile a CIM dump / real-data folder or a citi_bike build folder.  Round trips on
folders written here in the reference's formats; and, where the reference checkout / its built toy data happen to be
present (the build container), byte-for-byte agreement with the topologies the reference's own loaders produced."""
import json
import os

import numpy as np
import pytest

from maro_amd.data_lib import items_in_range, pick_ticks, read_binary, write_binary

REF_CIM = "/root/reference/tests/data/cim/case_data"
REF_CB_BUILD = "/tmp/oracle/home/.maro/data/citi_bike/.build"
REF_CB_TOPO = "/root/reference/maro/simulator/scenarios/citi_bike/topologies"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_binary_round_trip(tmp_path):
    p = str(tmp_path / "x.bin")
    ts = np.array([100, 160, 161, 400, 399, 1000], np.int64)
    write_binary(p, {"timestamp": ts, "a": np.arange(6), "b": np.arange(6) * 0.5, "c": np.arange(6) - 3},
                 {"timestamp": "i8", "a": "i", "b": "d", "c": "i2"}, raw_names={"timestamp": "tick"})
    hdr, rec = read_binary(p)
    assert hdr["item_count"] == 6 and hdr["starttime"] == 100 and hdr["endtime"] == 1000 and hdr["raw_names"]["timestamp"] == "tick"
    assert rec.dtype.names == ("timestamp", "a", "b", "c") and rec["c"].dtype == np.int16 and rec["b"].dtype == np.float64
    assert rec["timestamp"].tolist() == ts.tolist() and rec["b"].tolist() == [0, .5, 1, 1.5, 2, 2.5]
    # ItemTickPicker (binary_reader.py:80-112): 1 tick = 60 s from starttime; 399 arrives after 400 was handed out at tick 5
    assert pick_ticks(rec["timestamp"], 100, 20, "m").tolist() == [0, 1, 1, 5, -1, 15]
    assert pick_ticks(rec["timestamp"], 100, 10, "m").tolist() == [0, 1, 1, 5, -1, -1]
    assert items_in_range(rec, 100, 1, 5, time_unit="m")["a"].tolist() == [1, 2, 3, 4]


def _write_cim_folder(folder, binary):
    os.makedirs(folder)
    rows = lambda header, data: "\n".join([",".join(header)] + [",".join(str(x) for x in r) for r in data]) + "\n"  # noqa: E731
    open(os.path.join(folder, "misc.yml"), "w").write("container_volume: 1\ndsch_cost_factor: 0.05\nfuture_stop_number: 3\nload_cost_factor: 0.05\n"
                                                       "max_tick: 40\npast_stop_number: 4\nseed: 77\n")
    open(os.path.join(folder, "ports.csv"), "w").write(rows(
        ["index", "name", "capacity", "empty", "empty_return_buffer", "empty_return_buffer_noise", "full_return_buffer", "full_return_buffer_noise"],
        [[0, "A", 1000, 300, 1, 1, 1, 0], [1, "B", 900, 200, 2, 0, 1, 1], [2, "C", 800, 100, 1, 0, 2, 1]]))
    open(os.path.join(folder, "routes.csv"), "w").write(rows(["index", "name", "port_name", "port_index", "distance_to_next_port"],
                                                              [[0, "r0", "A", 0, 20], [0, "r0", "B", 1, 30], [0, "r0", "C", 2, 25], [1, "r1", "C", 2, 18], [1, "r1", "A", 0, 18]]))
    open(os.path.join(folder, "vessels.csv"), "w").write(rows(
        ["index", "name", "capacity", "route_name", "route_index", "start_port_name", "start_port_index", "sailing_speed", "sailing_speed_noise",
         "parking_duration", "parking_noise", "period", "empty"],
        [[0, "v0", 500, "r0", 0, "B", 1, 10, 1, 1, 0, 11, 5], [1, "v1", 400, "r1", 1, "C", 2, 9, 0, 2, 1, 8, 0]]))
    # stops: v0 starts at B (route r0 position 1): B, C, A, B...; v1 at C: C, A, C...
    stops = [(0, 0, 1, 1), (0, 4, 5, 2), (0, 8, 10, 0), (0, 13, 14, 1), (0, 18, 19, 2), (0, 22, 23, 0), (0, 26, 28, 1), (0, 31, 32, 2), (0, 35, 36, 0), (0, 39, 41, 1), (0, 44, 45, 2), (0, 48, 49, 0), (0, 52, 53, 1),
             (1, 0, 2, 2), (1, 4, 6, 0), (1, 8, 10, 2), (1, 12, 14, 0), (1, 16, 18, 2), (1, 20, 22, 0), (1, 24, 26, 2), (1, 28, 30, 0), (1, 32, 34, 2), (1, 36, 38, 0), (1, 40, 42, 2), (1, 44, 46, 0), (1, 48, 50, 2)]
    orders = [(t, s, d, 5 + (t * 7 + s * 3 + d) % 11) for t in range(40) for s, d in ((0, 1), (0, 2), (1, 2), (2, 0)) if (t + s + d) % 3]
    if binary:
        write_binary(os.path.join(folder, "stops.bin"), {"timestamp": [s[1] for s in stops], "leave_tick": [s[2] for s in stops], "port_index": [s[3] for s in stops],
                                                         "vessel_index": [s[0] for s in stops]}, {"timestamp": "i8", "leave_tick": "i", "port_index": "i", "vessel_index": "i"})
        write_binary(os.path.join(folder, "orders.bin"), {"timestamp": [o[0] for o in orders], "src_port_index": [o[1] for o in orders], "dest_port_index": [o[2] for o in orders],
                                                          "quantity": [o[3] for o in orders]}, {"timestamp": "i8", "src_port_index": "i", "dest_port_index": "i", "quantity": "i"})
    else:
        open(os.path.join(folder, "stops.csv"), "w").write(rows(["vessel_index", "port_index", "arrival_tick", "departure_tick"], [(s[0], s[3], s[1], s[2]) for s in stops]))
        open(os.path.join(folder, "orders.csv"), "w").write(rows(["tick", "source_port_index", "dest_port_index", "quantity"], orders))
    return stops, orders


def test_cim_real_data_folder_csv_and_binary_compile_to_the_same_topology(tmp_path):
    from maro_amd.cim.topology import load_data_folder
    stops, orders = _write_cim_folder(str(tmp_path / "csv"), False)
    _write_cim_folder(str(tmp_path / "bin"), True)
    a, b = load_data_folder(str(tmp_path / "csv"), name="x"), load_data_folder(str(tmp_path / "bin"), name="x")
    assert a.to_json() == b.to_json()
    assert a.data_mode == 2 and a.n_ports == 3 and a.n_vessels == 2 and a.data_max_tick == 40 and a.seed == 77
    assert a.target_offset.tolist() == [0, 2, 3, 4] and a.target_port.tolist() == [1, 2, 2, 0]
    assert a.fixed_n_stops.tolist() == [13, 13] and a.fixed_stops_arrival[1, :4].tolist() == [0, 4, 8, 12]
    want = np.zeros((40, 4), np.int32)
    for t, s, d, q in orders:
        want[t, [(0, 1), (0, 2), (1, 2), (2, 0)].index((s, d))] = q
    assert np.array_equal(a.fixed_orders, want)
    # vessel_period_without_noise of real data is computed: sum(parking + ceil(distance / speed)) over the route (:93-109)
    assert a.fixed_vessel_period.tolist() == [3 * 1 + 2 + 3 + 3, 2 * 2 + 2 + 2]


@pytest.mark.skipif(not os.path.isdir(REF_CIM), reason="reference fixtures not present (build container only)")
@pytest.mark.parametrize("folder,gold", [("real_folder_csv", "real_folder_csv"), ("real_folder_bin", "real_folder_bin")])
def test_cim_reference_fixture_folders_compile_to_the_committed_topologies(folder, gold):
    """tests/golden/topology_<gold>.json was written in round 1 from the reference's OWN loaders (cim_data_loader.py:360-450)."""
    from maro_amd.cim.topology import CimTopology, load_data_folder
    t = load_data_folder(os.path.join(REF_CIM, folder), name=gold)
    g = CimTopology.from_json(open(os.path.join(GOLDEN, f"topology_{gold}.json")).read())
    assert json.loads(t.to_json()) == json.loads(g.to_json())


def _write_citi_bike_build(folder, S=4, n_trips=500, seed=0):
    os.makedirs(folder)
    rng = np.random.RandomState(seed)
    t0 = 1546318800   # 2019-01-01 00:00 America/New_York, in UTC seconds
    ts = np.sort(t0 + rng.randint(0, 3 * 86400, n_trips))
    src = rng.randint(0, S, n_trips)
    dst = (src + rng.randint(1, S, n_trips)) % S
    write_binary(os.path.join(folder, "trips.bin"), {"timestamp": ts, "durations": rng.randint(2, 40, n_trips), "src_station": src, "dest_station": dst},
                 {"timestamp": "i8", "durations": "i", "src_station": "i", "dest_station": "i"})
    write_binary(os.path.join(folder, "KNYC_daily.bin"), {"timestamp": t0 + 86400 * np.arange(3) + 18000, "weather": [1, 2, 0], "temp": [3.6, -1.2, 10.9]},
                 {"timestamp": "i8", "weather": "i", "temp": "f"})
    with open(os.path.join(folder, "station_meta.csv"), "w") as fp:
        fp.write("station_index,capacity,init,station_id\n" + "".join(f"{i},{20 + i},{10 + i},{7000 + i}.0\n" for i in range(S)))
    d = rng.uniform(0.2, 3.0, (S, S))
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0.0)
    with open(os.path.join(folder, "distance_adj.csv"), "w") as fp:
        fp.write(",".join(str(i) for i in range(S)) + "\n" + "".join(",".join(repr(float(x)) for x in row) + "\n" for row in d))
    return ts, src, dst, d, t0


CB_CONF = {"time_zone": "America/New_York",
           "decision": {"extra_cost_mode": "source", "resolution": 20, "effective_time_mean": 20, "effective_time_std": 5, "supply_water_mark_ratio": 0.8,
                        "demand_water_mark_ratio": 0.2, "action_scope": {"low": 0, "high": 1, "filters": [{"type": "distance", "num": 3},
                                                                                                      {"type": "trip_window", "windows": 10, "num": 2}]}}}


def test_citi_bike_build_folder_loader(tmp_path):
    from maro_amd.citi_bike.data import load_build_folder
    ts, src, dst, d, t0 = _write_citi_bike_build(str(tmp_path / "b"))
    data = load_build_folder(CB_CONF, str(tmp_path / "b"), name="syn4")
    assert data.n_stations == 4 and data.capacity.tolist() == [20, 21, 22, 23] and data.init_bikes.tolist() == [10, 11, 12, 13] and data.station_id.tolist() == [7000, 7001, 7002, 7003]
    assert np.array_equal(data.trip_tick, ((ts - ts.min()) // 60).astype(np.int32)) and np.array_equal(data.trip_src, src) and np.array_equal(data.trip_dst, dst)
    assert np.allclose(data.distance, d) and data.filters == [dict(type=0, num=3, windows=0), dict(type=2, num=2, windows=10)]
    # calendar: ticks are minutes from the first trip, days are local (New York) dates; Jan 1st 2019 was a Tuesday
    assert data.day_weekday[:3].tolist() == [1, 2, 3] and data.day_weather[:3].tolist() == [1, 2, 0] and data.day_temperature[:3].tolist() == [3, -1, 10]
    assert data.day_holiday[:3].tolist() == [1, 0, 0]      # New Year's Day: the default calendar is holidays.US() restated
    assert not load_build_folder(CB_CONF, str(tmp_path / "b"), name="syn4", is_holiday=None).day_holiday.any()
    first_midnight = int(np.argmax(data.tick_day > 0))
    assert (ts.min() + 60 * first_midnight - t0) // 86400 == 1 and (ts.min() + 60 * (first_midnight - 1) - t0) // 86400 == 0


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_CB_BUILD, "toy.3s_4t")) or not os.path.isdir(REF_CB_TOPO), reason="built toy data not present (build container only)")
@pytest.mark.parametrize("name", ["toy.3s_4t", "toy.4s_4t", "toy.5s_6t"])
def test_citi_bike_packaged_topologies_are_what_the_native_loader_compiles(name):
    """The packaged maro_amd/citi_bike/topologies/<name>.npz were compiled in round 1 with the reference's BinaryReader /
    ItemTickPicker / WeatherTable; the native loader must produce the same arrays from the same build folder."""
    from maro_amd.citi_bike.data import load_build_folder, load_topology
    d, g = load_build_folder(os.path.join(REF_CB_TOPO, name, "config.yml"), os.path.join(REF_CB_BUILD, name), name=name, is_holiday=None), load_topology(name)
    for k in ("trip_tick", "trip_src", "trip_dst", "trip_duration", "capacity", "init_bikes", "station_id", "distance", "tick_day", "day_weekday",
              "day_holiday", "day_weather", "day_temperature"):
        assert np.array_equal(getattr(d, k), getattr(g, k)), k
    assert (d.resolution, d.time_mean, d.time_std, d.filters, d.extra_cost_mode) == (g.resolution, g.time_mean, g.time_std, g.filters, g.extra_cost_mode)


def test_compiled_real_data_folder_steps_identically_on_engine_and_oracle(tmp_path):
    """The topology compiled from the folder written above drives the device code (CPU wave emulator) and the oracle to the
    same decisions and metrics: the native loader's output is a valid data_mode-2 input of the engine."""
    from maro_amd.cim.topology import load_data_folder
    from oracle.cim_oracle import CimOracle, hash_policy_action
    from tests.emu.emu import EmuBackend
    _write_cim_folder(str(tmp_path / "bin"), True)
    topo = load_data_folder(str(tmp_path / "bin"), name="x")
    o = CimOracle(topo, durations=40)
    b = EmuBackend(topo, n_envs=2, durations=40, max_actions=1)
    b.reset(np.full(2, topo.seed, np.int64))
    dec, met, done = b.step()
    om, od, odone = o.step(None)
    k = 0
    while not odone:
        assert not done[0] and np.array_equal(dec[0], od) and np.array_equal(met[1], om), k
        a = hash_policy_action(5, k, od)
        acts = np.zeros((2, 1, 4), np.int32)
        acts[:, 0] = a
        dec, met, done = b.step(acts, np.ones(2, np.int32))
        om, od, odone = o.step([a])
        k += 1
    assert done.all() and np.array_equal(met[0], om) and k > 5


def test_reader_semantics_match_the_reference_readers():
    """pick_ticks / items_in_range against what the REAL BinaryReader yielded (oracle/gen_golden_data_lib.py: sorted, duplicated
    and out-of-order timestamps, units s / m / h, windows that start before / at the first record)."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_lib_reader_kat.json")))
    assert len(gold["cases"]) >= 10
    for c in gold["cases"]:
        ts = np.array(c["timestamps"], np.int64)
        assert pick_ticks(ts, c["starttime"], c["n_ticks"], c["unit"]).tolist() == c["ticks"], c
        rec = np.zeros(len(ts), dtype=[("timestamp", "<i8"), ("idx", "<i4")])
        rec["timestamp"], rec["idx"] = ts, np.arange(len(ts))
        lo, hi = c["range"]
        assert items_in_range(rec, c["starttime"], lo, hi, time_unit=c["unit"])["idx"].tolist() == c["range_idx"], c
