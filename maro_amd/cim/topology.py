"""Host-side topology compiler for the CIM rollout engine.

Turns a MARO CIM topology (the ``config.yml`` schema consumed by the reference's
``maro/data_lib/cim/parsers.py:14-211`` and ``cim_data_generator.py:118-205``) into the flat
arrays of ``mrx_cim_topology`` (``include/maro_amd.h``).  Only *parsing* happens here; every
stochastic step (route unrolling, order proportion noise, RNG seeding) runs on the device.

Index conventions follow the reference: port / vessel / route indices are yml order
(``parsers.py:27-52, 109-131, 151-156``).
"""
from __future__ import annotations

import ctypes
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

_PKG_TOPOLOGY_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "topologies")

_F64 = ("order_dist", "empty_return_base", "empty_return_noise", "full_return_base", "full_return_noise",
        "source_base", "source_noise", "target_base", "target_noise", "route_dist", "vessel_speed",
        "vessel_speed_noise", "vessel_duration", "vessel_duration_noise")
_I32 = ("port_capacity", "port_init_empty", "target_offset", "target_port", "route_offset", "route_port",
        "vessel_capacity", "vessel_init_empty", "vessel_route", "vessel_start_offset")
_FIXED_I32 = ("fixed_n_stops", "fixed_stops_arrival", "fixed_stops_leave", "fixed_vessel_period", "fixed_order_prop",
              "fixed_orders")
_FIXED_SCALARS = ("data_mode", "data_max_tick", "fixed_max_stops")
_SCALARS = ("n_ports", "n_vessels", "n_routes", "n_targets", "n_route_points", "past_stop_number",
            "future_stop_number", "container_volume", "total_containers", "order_mode", "seed", "period",
            "sample_noise")


class MrxCimTopology(ctypes.Structure):
    """ctypes mirror of ``struct mrx_cim_topology`` (include/maro_amd.h)."""

    _fields_ = (
        [(n, ctypes.c_int32) for n in ("n_ports", "n_vessels", "n_routes", "n_targets", "n_route_points",
                                       "past_stop_number", "future_stop_number", "container_volume",
                                       "total_containers", "order_mode")]
        + [("seed", ctypes.c_int64), ("period", ctypes.c_int32), ("sample_noise", ctypes.c_double),
           ("order_dist", ctypes.POINTER(ctypes.c_double))]
        + [("port_capacity", ctypes.POINTER(ctypes.c_int32)), ("port_init_empty", ctypes.POINTER(ctypes.c_int32))]
        + [(n, ctypes.POINTER(ctypes.c_double)) for n in ("empty_return_base", "empty_return_noise",
                                                           "full_return_base", "full_return_noise",
                                                           "source_base", "source_noise")]
        + [("target_offset", ctypes.POINTER(ctypes.c_int32)), ("target_port", ctypes.POINTER(ctypes.c_int32)),
           ("target_base", ctypes.POINTER(ctypes.c_double)), ("target_noise", ctypes.POINTER(ctypes.c_double))]
        + [("route_offset", ctypes.POINTER(ctypes.c_int32)), ("route_port", ctypes.POINTER(ctypes.c_int32)),
           ("route_dist", ctypes.POINTER(ctypes.c_double))]
        + [(n, ctypes.POINTER(ctypes.c_int32)) for n in ("vessel_capacity", "vessel_init_empty", "vessel_route",
                                                          "vessel_start_offset")]
        + [(n, ctypes.POINTER(ctypes.c_double)) for n in ("vessel_speed", "vessel_speed_noise",
                                                           "vessel_duration", "vessel_duration_noise")]
        + [(n, ctypes.c_int32) for n in ("data_mode", "data_max_tick", "fixed_max_stops")]
        + [(n, ctypes.POINTER(ctypes.c_int32)) for n in _FIXED_I32]
    )


@dataclass
class CimTopology:
    """Flat, array-form CIM topology (one instance is shared by every env of a batch)."""

    name: str
    n_ports: int
    n_vessels: int
    n_routes: int
    n_targets: int
    n_route_points: int
    past_stop_number: int
    future_stop_number: int
    container_volume: int
    total_containers: int
    order_mode: int
    seed: int
    period: int
    sample_noise: float
    order_dist: np.ndarray
    port_capacity: np.ndarray
    port_init_empty: np.ndarray
    empty_return_base: np.ndarray
    empty_return_noise: np.ndarray
    full_return_base: np.ndarray
    full_return_noise: np.ndarray
    source_base: np.ndarray
    source_noise: np.ndarray
    target_offset: np.ndarray
    target_port: np.ndarray
    target_base: np.ndarray
    target_noise: np.ndarray
    route_offset: np.ndarray
    route_port: np.ndarray
    route_dist: np.ndarray
    vessel_capacity: np.ndarray
    vessel_init_empty: np.ndarray
    vessel_route: np.ndarray
    vessel_start_offset: np.ndarray
    vessel_speed: np.ndarray
    vessel_speed_noise: np.ndarray
    vessel_duration: np.ndarray
    vessel_duration_noise: np.ndarray
    port_names: List[str] = field(default_factory=list)
    vessel_names: List[str] = field(default_factory=list)
    route_names: List[str] = field(default_factory=list)
    load_cost_factor: float = 0.0
    dsch_cost_factor: float = 0.0
    raw_config: dict = field(default_factory=dict, repr=False)
    # data read from files (dump folder / real data files) instead of generated at reset — include/maro_amd.h
    data_mode: int = 0
    data_max_tick: int = 0
    fixed_max_stops: int = 0
    fixed_n_stops: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    fixed_stops_arrival: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    fixed_stops_leave: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    fixed_vessel_period: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    fixed_order_prop: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    fixed_orders: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    # ---- reference-compatible views -------------------------------------------------
    @property
    def port_mapping(self) -> Dict[str, int]:
        return {n: i for i, n in enumerate(self.port_names)}

    @property
    def vessel_mapping(self) -> Dict[str, int]:
        return {n: i for i, n in enumerate(self.vessel_names)}

    def route_length(self, vessel_idx: int) -> int:
        r = int(self.vessel_route[vessel_idx])
        return int(self.route_offset[r + 1] - self.route_offset[r])

    # ---- C ABI ---------------------------------------------------------------------
    def c_struct(self) -> MrxCimTopology:
        """Build the ctypes struct; array storage is kept alive on the returned object."""
        s = MrxCimTopology()
        keep = []
        for n in _SCALARS:
            setattr(s, n, getattr(self, n))
        for n in _F64:
            a = np.ascontiguousarray(getattr(self, n), dtype=np.float64)
            keep.append(a)
            setattr(s, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        for n in _I32:
            a = np.ascontiguousarray(getattr(self, n), dtype=np.int32)
            keep.append(a)
            setattr(s, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        for n in _FIXED_SCALARS:
            setattr(s, n, int(getattr(self, n)))
        for n in _FIXED_I32:
            a = np.ascontiguousarray(getattr(self, n), dtype=np.int32).reshape(-1)
            keep.append(a)
            setattr(s, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if a.size else None)
        s._keepalive = keep
        return s

    # ---- (de)serialisation of the packaged, pre-compiled form -----------------------
    def to_json(self) -> str:
        d = {n: getattr(self, n) for n in _SCALARS}
        d["name"] = self.name
        # repr() round-trips float64 exactly
        for n in _F64:
            d[n] = [repr(float(x)) for x in getattr(self, n)]
        for n in _I32:
            d[n] = [int(x) for x in getattr(self, n)]
        d["sample_noise"] = repr(float(self.sample_noise))
        d.update(port_names=self.port_names, vessel_names=self.vessel_names, route_names=self.route_names,
                 load_cost_factor=self.load_cost_factor, dsch_cost_factor=self.dsch_cost_factor)
        if self.data_mode:
            for n in _FIXED_SCALARS:
                d[n] = int(getattr(self, n))
            for n in _FIXED_I32:
                d[n] = np.asarray(getattr(self, n), np.int32).reshape(-1).tolist()
        return json.dumps(d, separators=(",", ":"))

    @staticmethod
    def from_json(text: str) -> "CimTopology":
        d = json.loads(text)
        kw = dict(d)
        kw["sample_noise"] = float(d["sample_noise"])
        for n in _F64:
            kw[n] = np.array([float(x) for x in d[n]], dtype=np.float64)
        for n in _I32:
            kw[n] = np.array(d[n], dtype=np.int32)
        for n in _FIXED_I32:
            if n in d:
                kw[n] = np.array(d[n], dtype=np.int32)
        return CimTopology(**kw)


def _is_intlike(x) -> bool:
    return float(x) == int(x)


def parse_config(conf: dict, name: str = "custom") -> CimTopology:
    """Flatten a CIM ``config.yml`` dict (schema: reference parsers.py / cim_data_generator.py:141-205)."""
    total_containers = conf["total_containers"]
    past_n, future_n = conf["stop_number"]
    volume = conf["container_volumes"][0]
    if not _is_intlike(volume):
        raise ValueError("container_volumes[0] must be an integer")

    # container_usage_proportion -> one period of the order ratio (parsers.py:70-91)
    cup = conf["container_usage_proportion"]
    period = int(cup["period"])
    nodes = [(x, y) for x, y in cup["sample_nodes"]]
    if nodes[0][0] != 0:
        nodes.insert(0, (0, 0))
    if nodes[-1][0] != period - 1:
        nodes.append((period - 1, 0))
    order_dist = np.interp(list(range(period)), [n[0] for n in nodes], [n[1] for n in nodes]).astype(np.float64)

    # ports (parsers.py:134-211)
    ports = conf["ports"]
    total_ratio = sum(p["initial_container_proportion"] for p in ports.values())
    assert round(total_ratio, 7) == 1, "initial_container_proportion must sum to 1 (parsers.py:145-146)"
    port_names = list(ports.keys())
    pmap = {n: i for i, n in enumerate(port_names)}
    t_off, t_port, t_base, t_noise = [0], [], [], []
    for info in ports.values():
        for tgt_name, tc in (info["order_distribution"].get("targets") or {}).items():
            t_port.append(pmap[tgt_name])
            t_base.append(tc["proportion"])
            t_noise.append(tc["noise"])
        t_off.append(len(t_port))

    # routes (parsers.py:109-131)
    routes = conf["routes"]
    route_names = list(routes.keys())
    rmap = {n: i for i, n in enumerate(route_names)}
    r_off, r_port, r_dist = [0], [], []
    for pts in routes.values():
        for pt in pts:
            r_port.append(pmap[pt["port_name"]])
            r_dist.append(pt["distance_to_next_port"])
        r_off.append(len(r_port))

    # vessels (parsers.py:14-54); start offset as in vessel_future_stops_prediction.py:39-47
    vessels = conf["vessels"]
    vessel_names = list(vessels.keys())
    v_route, v_start = [], []
    for v in vessels.values():
        r = rmap[v["route"]["route_name"]]
        names = [pt["port_name"] for pt in routes[v["route"]["route_name"]]]
        v_route.append(r)
        v_start.append(names.index(v["route"]["initial_port_name"]))
        if not _is_intlike(v["parking"]["duration"]):
            raise ValueError("parking.duration must be an integer number of ticks")

    mode = conf["order_generate_mode"]
    f64 = lambda xs: np.array([float(x) for x in xs], dtype=np.float64)  # noqa: E731
    i32 = lambda xs: np.array([int(x) for x in xs], dtype=np.int32)  # noqa: E731
    return CimTopology(
        name=name,
        n_ports=len(ports), n_vessels=len(vessels), n_routes=len(routes),
        n_targets=len(t_port), n_route_points=len(r_port),
        past_stop_number=int(past_n), future_stop_number=int(future_n),
        container_volume=int(volume), total_containers=int(total_containers),
        order_mode={"fixed": 0, "unfixed": 1}[mode], seed=int(conf["seed"]),
        period=period, sample_noise=float(cup["sample_noise"]), order_dist=order_dist,
        port_capacity=i32(p["capacity"] for p in ports.values()),
        port_init_empty=i32(int(p["initial_container_proportion"] * total_containers) for p in ports.values()),
        empty_return_base=f64(p["empty_return"]["buffer_ticks"] for p in ports.values()),
        empty_return_noise=f64(p["empty_return"]["noise"] for p in ports.values()),
        full_return_base=f64(p["full_return"]["buffer_ticks"] for p in ports.values()),
        full_return_noise=f64(p["full_return"]["noise"] for p in ports.values()),
        source_base=f64(p["order_distribution"]["source"]["proportion"] for p in ports.values()),
        source_noise=f64(p["order_distribution"]["source"]["noise"] for p in ports.values()),
        target_offset=i32(t_off), target_port=i32(t_port), target_base=f64(t_base), target_noise=f64(t_noise),
        route_offset=i32(r_off), route_port=i32(r_port), route_dist=f64(r_dist),
        vessel_capacity=i32(v["capacity"] for v in vessels.values()),
        vessel_init_empty=i32(v.get("empty", 0) for v in vessels.values()),
        vessel_route=i32(v_route), vessel_start_offset=i32(v_start),
        vessel_speed=f64(v["sailing"]["speed"] for v in vessels.values()),
        vessel_speed_noise=f64(v["sailing"]["noise"] for v in vessels.values()),
        vessel_duration=f64(v["parking"]["duration"] for v in vessels.values()),
        vessel_duration_noise=f64(v["parking"]["noise"] for v in vessels.values()),
        port_names=port_names, vessel_names=vessel_names, route_names=route_names,
        load_cost_factor=float(conf.get("load_cost_factor", 0.0)),
        dsch_cost_factor=float(conf.get("dsch_cost_factor", 0.0)),
        raw_config=conf,
    )


def available_topologies() -> List[str]:
    if not os.path.isdir(_PKG_TOPOLOGY_DIR):
        return []
    return sorted(f[:-5] for f in os.listdir(_PKG_TOPOLOGY_DIR) if f.endswith(".json"))


def load_topology(topology: str) -> CimTopology:
    """Resolve a topology the way the reference does (abs_business_engine.py:131-164): a built-in
    name, or a filesystem path to a folder holding ``config.yml`` (or to the yml itself)."""
    if os.path.exists(topology):
        import yaml

        path = os.path.join(topology, "config.yml") if os.path.isdir(topology) else topology
        with open(path) as fp:
            conf = yaml.safe_load(fp)
        return parse_config(conf, name=os.path.basename(os.path.normpath(topology)))
    packaged = os.path.join(_PKG_TOPOLOGY_DIR, topology + ".json")
    if os.path.exists(packaged):
        with open(packaged) as fp:
            return CimTopology.from_json(fp.read())
    raise FileNotFoundError(f"unknown CIM topology {topology!r}; built-ins: {available_topologies()}")
