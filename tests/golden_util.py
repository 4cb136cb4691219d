"""Helpers shared by the golden-vector tests (oracle on CPU, HIP engine on GPU)."""
import json
import os

import numpy as np

from maro_amd.cim.topology import CimTopology, load_topology

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PORT_ATTRS = ["capacity", "empty", "full", "on_shipper", "on_consignee", "shortage", "acc_shortage", "booking",
              "acc_booking", "fulfillment", "acc_fulfillment", "transfer_cost"]
VESSEL_ATTRS = ["capacity", "empty", "full", "remaining_space", "early_discharge", "is_parking", "loc_port_idx",
                "route_idx", "last_loc_idx", "next_loc_idx", "past_stop_list", "past_stop_tick_list",
                "future_stop_list", "future_stop_tick_list"]
MATRIX_ATTRS = ["full_on_ports", "full_on_vessels", "vessel_plans"]


def golden_cases():
    return sorted(f[4:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("cim_") and f.endswith(".npz"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"cim_{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def case_topology(meta) -> CimTopology:
    topo = meta["topology"]
    if topo.startswith("#"):
        from maro_amd.cim.topology import parse_config
        from tests.test_emu_synthetic import VARIANTS
        import copy
        return parse_config(copy.deepcopy(VARIANTS[topo[1:]]), name="syn_" + topo[1:])
    if topo.startswith("="):   # "=<name>:<folder>": a dump / real data folder compiled by tools/import_maro_cim_data.py
        with open(os.path.join(GOLDEN_DIR, f"topology_{topo[1:].split(':')[0]}.json")) as fp:
            return CimTopology.from_json(fp.read())
    if topo.startswith("@"):
        with open(os.path.join(GOLDEN_DIR, "topology_case_config_folder.json")) as fp:
            return CimTopology.from_json(fp.read())
    return load_topology(topo)


def segment_actions(z, seg):
    """Recorded action lists per decision: list of [(vessel, port, qty, type), ...]."""
    acts = z[f"seg{seg}/actions"]
    return [[tuple(int(x) for x in a) for a in row if a[0] >= 0] for row in acts]


def joint_golden_cases():
    return sorted(f[9:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("cimjoint_") and f.endswith(".npz"))


def load_joint_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"cimjoint_{name}.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


class JointPayloadCache:
    """The reference re-yields the SAME DecisionEvent objects for events that stayed pending in
    JointWithSequentialAction mode, and a DecisionEvent caches action_scope / early_discharge at its first read
    (cim/common.py:107-123).  Rows coming from an engine (always evaluated on the live state) go through this cache
    to become what an agent of the reference sees."""

    def __init__(self):
        self._seen = {}

    def __call__(self, rows):
        out = rows.copy()
        for i in range(len(rows)):
            if rows[i, 7] == 1:
                key = (int(rows[i, 0]), int(rows[i, 2]))
                if key in self._seen:
                    out[i] = self._seen[key]
                else:
                    self._seen[key] = rows[i].copy()
        return out


def replay_joint_case(make_env, name):
    """make_env(topology, kwargs, decision_mode) -> object with step_joint(actions, n_answered) -> (metrics, rows[V,8], done),
    .tick, .frame_indices(), .query(node, ticks, nodes, attrs)."""
    z, meta = load_joint_case(name)
    env = make_env(case_topology(meta), meta["kwargs"], meta["decision_mode"])
    cache = JointPayloadCache()
    gd, gm, ga, gn = z["decisions"], z["metrics"], z["actions"], z["n_answered"]
    met, rows, done = env.step_joint(None, 0)
    i = 0
    while not done:
        rows = cache(rows)
        assert np.array_equal(rows, gd[i]), (name, i, rows[rows[:, 7] == 1], gd[i][gd[i][:, 7] == 1])
        assert np.array_equal(met, gm[i]), (name, i, met, gm[i])
        k = int(gn[i])
        met, rows, done = env.step_joint([tuple(int(x) for x in a) for a in ga[i][:k]], k)
        i += 1
    assert i == len(gd)
    assert np.array_equal(met, z["final_metrics"])
    assert env.tick == int(z["final_tick"][0])
    assert env.frame_indices() == z["frame_indices"].tolist()
    assert np.array_equal(env.query("ports", [], [], PORT_ATTRS), z["snap_ports"])
    assert np.array_equal(env.query("vessels", [], [], VESSEL_ATTRS), z["snap_vessels"])
    assert np.array_equal(env.query("matrices", [], [], MATRIX_ATTRS), z["snap_matrices"])
