"""citi_bike: the pure-Python oracle pinned against vectors produced by the real reference on the packaged
toy.3s_4t data (oracle/gen_golden_citi_bike.py)."""
import json
import os

import numpy as np
import pytest

from maro_amd.citi_bike.data import load_topology
from oracle.citi_bike_oracle import STATION_ATTRS, CitiBikeOracle, draw_transfer_times

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("cb_") and f.endswith(".npz"))


def load(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def replay_citi_bike(make_env, case):
    z, meta = load(case)
    data = load_topology(meta["topology"])
    env = make_env(data, meta["kwargs"], draw_transfer_times(data, meta["np_seed"], max(4096, len(z["actions"]) + 8)))
    gd, gs, gm, ga = z["decisions"], z["scopes"], z["metrics"], z["actions"]
    m, de, done = env.step(None)
    i = 0
    while not done:
        assert [de["tick"], de["station_idx"], de["type"], de["frame_index"], len(de["action_scope"])] == gd[i].tolist(), (case, i, de)
        assert [list(x) for x in de["action_scope"]] == gs[i][: gd[i][4]].tolist(), (case, i, de["action_scope"], gs[i])
        assert [m["trip_requirements"], m["bike_shortage"], m["operation_number"]] == gm[i].tolist(), (case, i)
        a = ga[i]
        m, de, done = env.step([tuple(int(x) for x in a)] if a[0] >= 0 else None)
        i += 1
    assert i == len(gd)
    assert [m["trip_requirements"], m["bike_shortage"], m["operation_number"]] == z["final_metrics"].tolist()
    assert env.tick == int(z["final_tick"][0])
    assert env.frame_indices() == z["frame_indices"].tolist()
    assert np.array_equal(env.query("stations", [], [], STATION_ATTRS), z["snap_stations"])
    assert np.array_equal(env.query("matrices", [], [], ["trips_adj"]), z["snap_matrices"])


@pytest.mark.parametrize("case", CASES)
def test_citi_bike_oracle_reproduces_reference(case):
    replay_citi_bike(lambda data, kw, tt: CitiBikeOracle(data, transfer_times=tt, **kw), case)
