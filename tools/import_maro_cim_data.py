#!/usr/bin/env python3
"""Compile a MARO CIM *dump folder* (data_from_dumps) or *real data folder* (data_from_files) into the engine's
packaged topology JSON (maro_amd/cim/topology.py, data_mode 1 / 2).

The on-disk formats (csv + MARO binary) are read with the reference's own loaders
(maro/data_lib/cim/cim_data_loader.py:360-450), so this tool needs a MARO checkout; the JSON it writes does not.

    python tools/import_maro_cim_data.py --maro /tmp/oracle/maro_src --folder <dump or real folder> --name my_data --out x.json
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_collection(dc, name, real):
    sys.path.insert(0, REPO)
    from maro_amd.cim.topology import CimTopology

    ports, vessels, routes = dc.port_settings, dc.vessel_settings, dc.routes
    P, V = len(ports), len(vessels)
    pmap, rmap = dc.port_mapping, dc.route_mapping
    f64 = lambda xs: np.array([float(x) for x in xs], dtype=np.float64)  # noqa: E731
    i32 = lambda xs: np.array([int(x) for x in xs], dtype=np.int32)  # noqa: E731
    r_off, r_port, r_dist = [0], [], []
    for pts in routes:
        for pt in pts:
            r_port.append(pmap[pt.port_name])
            r_dist.append(pt.distance_to_next_port)
        r_off.append(len(r_port))
    v_route, v_start = [], []
    for v in vessels:
        r = rmap[v.route_name]
        v_route.append(r)
        v_start.append([pt.port_name for pt in routes[r]].index(v.start_port_name))
    # stops: (arrival, leave) per vessel; the port of stop k must follow the route from the start position
    n_stops = [len(ss) for ss in dc.vessel_stops]
    smax = max(n_stops)
    arr, lea = np.zeros((V, smax), np.int32), np.zeros((V, smax), np.int32)
    for v, ss in enumerate(dc.vessel_stops):
        L = r_off[v_route[v] + 1] - r_off[v_route[v]]
        for k, s in enumerate(ss):
            want = r_port[r_off[v_route[v]] + (v_start[v] + k) % L]
            if s.port_idx != want:
                raise ValueError(f"vessel {v} stop {k} is at port {s.port_idx}, its route says {want}: not representable")
            arr[v, k], lea[v, k] = s.arrival_tick, s.leave_tick
    T = int(dc.max_tick)
    kw = {}
    if real:
        # the pair universe and its order come from the order file: within a tick the reference handles orders in file
        # order, which matters per source port (sequential use of `empty`) and for the buffer-tick draw order
        seq = {}
        for t in sorted(dc.orders):
            seen = set()
            for rank, o in enumerate(dc.orders[t]):
                key = (o.src_port_idx, o.dest_port_idx)
                if key in seen:
                    raise ValueError(f"tick {t}: two orders for the same (src, dst) pair are not representable")
                seen.add(key)
                seq.setdefault(key, []).append((t, rank))
        pairs = sorted(seq)   # CSR order: by source port, then destination
        for t in dc.orders:   # file order inside a tick must agree with the CSR order
            ks = [pairs.index((o.src_port_idx, o.dest_port_idx)) for o in dc.orders[t]]
            if ks != sorted(ks):
                raise ValueError(f"tick {t}: orders are not listed by (source, destination) port index: not representable")
        t_off = [0]
        for p in range(P):
            t_off.append(t_off[-1] + sum(1 for s, _ in pairs if s == p))
        orders = np.zeros((T, len(pairs)), np.int32)
        for t, lst in dc.orders.items():
            if t < T:
                for o in lst:
                    orders[t, pairs.index((o.src_port_idx, o.dest_port_idx))] = o.quantity
        kw.update(target_offset=i32(t_off), target_port=i32(d for _, d in pairs), target_base=f64([0] * len(pairs)),
                  target_noise=f64([0] * len(pairs)), source_base=f64([0] * P), source_noise=f64([0] * P),
                  fixed_orders=orders, fixed_order_prop=np.zeros(0, np.int32), total_containers=int(sum(p.empty for p in ports)),
                  order_mode=0, data_mode=2)
    else:
        t_off, t_port, t_base, t_noise = [0], [], [], []
        for p in ports:
            for tp in (p.target_proportions or []):
                t_port.append(tp.index); t_base.append(tp.base); t_noise.append(tp.noise)
            t_off.append(len(t_port))
        kw.update(target_offset=i32(t_off), target_port=i32(t_port), target_base=f64(t_base), target_noise=f64(t_noise),
                  source_base=f64(p.source_proportion.base for p in ports), source_noise=f64(p.source_proportion.noise for p in ports),
                  fixed_order_prop=np.asarray(dc.order_proportion, np.int32)[:T], fixed_orders=np.zeros(0, np.int32),
                  total_containers=int(dc.total_containers), order_mode={"fixed": 0, "unfixed": 1}[dc.order_mode.value], data_mode=1)
    return CimTopology(
        name=name, n_ports=P, n_vessels=V, n_routes=len(routes), n_targets=len(kw["target_port"]), n_route_points=len(r_port),
        past_stop_number=int(dc.past_stop_number), future_stop_number=int(dc.future_stop_number),
        container_volume=int(dc.container_volume), seed=int(dc.seed), period=1, sample_noise=0.0, order_dist=f64([0]),
        port_capacity=i32(p.capacity for p in ports), port_init_empty=i32(p.empty for p in ports),
        empty_return_base=f64(p.empty_return_buffer.base for p in ports), empty_return_noise=f64(p.empty_return_buffer.noise for p in ports),
        full_return_base=f64(p.full_return_buffer.base for p in ports), full_return_noise=f64(p.full_return_buffer.noise for p in ports),
        route_offset=i32(r_off), route_port=i32(r_port), route_dist=f64(r_dist),
        vessel_capacity=i32(v.capacity for v in vessels), vessel_init_empty=i32(v.empty for v in vessels),
        vessel_route=i32(v_route), vessel_start_offset=i32(v_start),
        vessel_speed=f64(v.sailing_speed for v in vessels), vessel_speed_noise=f64(v.sailing_noise for v in vessels),
        vessel_duration=f64(v.parking_duration for v in vessels), vessel_duration_noise=f64(v.parking_noise for v in vessels),
        port_names=[p.name for p in ports], vessel_names=[v.name for v in vessels],
        route_names=[n for n, _ in sorted(rmap.items(), key=lambda kv: kv[1])],
        load_cost_factor=float(dc.load_cost_factor), dsch_cost_factor=float(dc.dsch_cost_factor),
        data_max_tick=T, fixed_max_stops=smax, fixed_n_stops=i32(n_stops), fixed_stops_arrival=arr, fixed_stops_leave=lea,
        fixed_vessel_period=i32(dc.vessel_period_without_noise), **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--folder", required=True)
    ap.add_argument("--name")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    sys.path.insert(0, args.maro)
    from maro.data_lib.cim.cim_data_loader import load_from_folder, load_real_data_from_folder
    real = not os.path.exists(os.path.join(args.folder, "order_proportion.csv"))
    dc = load_real_data_from_folder(args.folder) if real else load_from_folder(args.folder)
    topo = compile_collection(dc, args.name or os.path.basename(os.path.normpath(args.folder)), real)
    with open(args.out, "w") as fp:
        fp.write(topo.to_json())
    print(f"{topo.name}: data_mode {topo.data_mode}, {topo.n_ports} ports, {topo.n_vessels} vessels, {topo.n_targets} order pairs, "
          f"max_tick {topo.data_max_tick}, <= {topo.fixed_max_stops} stops/vessel -> {args.out}")


if __name__ == "__main__":
    main()
