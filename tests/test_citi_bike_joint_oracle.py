"""citi_bike, Joint / JointWithSequentialAction decision modes (core.py:354-366): the oracle pinned against vectors produced by
the real reference (oracle/gen_golden_citi_bike_joint.py); `replay_citi_bike_joint` is shared with the device-code tests."""
import json
import os

import numpy as np
import pytest

from maro_amd.citi_bike.data import load_topology
from oracle.citi_bike_oracle import STATION_ATTRS, CitiBikeOracle, draw_transfer_times

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
JOINT_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("cbjoint_") and f.endswith(".npz"))


def replay_citi_bike_joint(make_env, case):
    """make_env(data, kwargs, transfer_times, decision_mode) -> env with step_joint(actions_per_event) -> (metrics, [decision dicts], done)."""
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    data = load_topology(meta["topology"])
    env = make_env(data, meta["kwargs"], draw_transfer_times(data, meta["np_seed"], max(4096, len(z["actions"]) + 8)), meta["decision_mode"])
    gd, gs, ga = z["decisions"], z["scopes"], z["actions"]
    m, des, done = env.step_joint(None)
    row = 0
    for step, (n_ev, n_ans) in enumerate(zip(z["n_events"].tolist(), z["n_answered"].tolist())):
        assert not done and len(des) == n_ev, (case, step, len(des or []), n_ev)
        assert [m["trip_requirements"], m["bike_shortage"], m["operation_number"]] == z["metrics"][step].tolist(), (case, step)
        acts = []
        for i, de in enumerate(des):
            assert [de["tick"], de["station_idx"], de["type"], de["frame_index"], len(de["action_scope"])] == gd[row + i].tolist(), (case, step, i, de)
            assert [list(x) for x in de["action_scope"]] == gs[row + i][: gd[row + i][4]].tolist(), (case, step, i, de["action_scope"])
            a = ga[row + i]
            acts.append([tuple(int(x) for x in a)] if a[0] >= 0 else None)
        row += n_ev
        m, des, done = env.step_joint(acts[:n_ans])
    assert done and row == len(gd)
    assert [m["trip_requirements"], m["bike_shortage"], m["operation_number"]] == z["final_metrics"].tolist()
    assert env.tick == int(z["final_tick"][0])
    assert env.frame_indices() == z["frame_indices"].tolist()
    assert np.array_equal(env.query("stations", [], [], STATION_ATTRS), z["snap_stations"])


class _JointOracle(CitiBikeOracle):
    def __init__(self, *a, decision_mode=1, **k):
        super().__init__(*a, **k)
        self._mode = decision_mode

    def step_joint(self, actions_per_event=None):
        return super().step_joint(actions_per_event, self._mode)


@pytest.mark.parametrize("case", JOINT_CASES)
def test_citi_bike_oracle_reproduces_reference_joint_modes(case):
    replay_citi_bike_joint(lambda data, kw, tt, mode: _JointOracle(data, transfer_times=tt, decision_mode=mode, **kw), case)
