"""The CPU oracle (oracle/cim_oracle.c) pinned against vectors produced by the REAL reference
(oracle/gen_golden.py) and against the reference's own known answers."""
import numpy as np
import pytest

from oracle.cim_oracle import CimOracle, mt_selftest
from tests.golden_util import (MATRIX_ATTRS, PORT_ATTRS, VESSEL_ATTRS, case_topology, golden_cases, joint_golden_cases, load_case,
                               replay_joint_case, segment_actions)


def test_mt19937_matches_cpython_random():
    import random

    for seed in (0, 1, 4096, 4099, 2**31 - 1, 2**32 + 5, 123456789012):
        r = random.Random(seed)
        exp_r = [r.random() for _ in range(700)]
        exp_b = [r.randint(0, 4095) for _ in range(700)]
        got_r, got_b = mt_selftest(seed, 700)
        assert got_r.tolist() == exp_r
        assert got_b.tolist() == exp_b


def replay_case(make_env, name):
    z, meta = load_case(name)
    env = make_env(case_topology(meta), meta["kwargs"])
    seg = 0
    pos = {}
    light = f"seg0/snap_ports" not in z

    def check_data(seg):
        tag = f"seg{seg}"
        arr, lea, prt = z[f"{tag}/stops_arrival"], z[f"{tag}/stops_leave"], z[f"{tag}/stops_port"]
        for v in range(arr.shape[0]):
            a, l, p = env.stops(v)
            n = int((arr[v] >= 0).sum())
            assert len(a) == n, (name, tag, v)
            assert np.array_equal(a, arr[v, :n]) and np.array_equal(l, lea[v, :n]) and np.array_equal(p, prt[v, :n])
        if len(z[f"{tag}/order_proportion"]):  # real-data collections have no order proportion (orders come from a file)
            assert np.array_equal(env.order_proportion(), z[f"{tag}/order_proportion"])
        assert np.array_equal(env.vessel_period(), z[f"{tag}/vessel_period"])
        assert env.data_seed == int(z[f"{tag}/data_seed"][0])

    check_data(0)
    started = False
    done = False
    met = dec = None
    for op in meta["script"]:
        if op[0] == "set_seed":
            env.set_seed(op[1])
        elif op[0] == "reset":
            env.reset(keep_seed=op[1])
            seg += 1
            started = done = False
            check_data(seg)
        else:
            tag = f"seg{seg}"
            gd, gm = z[f"{tag}/decisions"], z[f"{tag}/metrics"]
            acts = segment_actions(z, seg)
            i = pos.get(seg, 0)
            max_steps = op[2]
            n = 0
            if not started:
                met, dec, done = env.step(None)
                started = True
            while not done and (max_steps is None or n < max_steps):
                assert np.array_equal(dec, gd[i]), (name, tag, i, dec, gd[i])
                assert np.array_equal(met, gm[i]), (name, tag, i, met, gm[i])
                met, dec, done = env.step(acts[i])
                i += 1
                n += 1
            pos[seg] = i
            assert int(done) == int(z[f"{tag}/done"][0])
            assert env.tick == int(z[f"{tag}/final_tick"][0])
            if done:
                assert np.array_equal(met, z[f"{tag}/final_metrics"])
                assert i == len(gd)
            if f"{tag}/final_frame_index" in z:
                fi = [int(z[f"{tag}/final_frame_index"][0])]
                assert np.array_equal(env.query("ports", fi, [], PORT_ATTRS), z[f"{tag}/final_ports"])
                assert np.array_equal(env.query("vessels", fi, [], VESSEL_ATTRS), z[f"{tag}/final_vessels"])
                assert np.array_equal(env.query("matrices", fi, [], MATRIX_ATTRS), z[f"{tag}/final_matrices"])
            if not light:
                assert env.frame_indices() == z[f"{tag}/frame_indices"].tolist()
                assert np.array_equal(env.query("ports", [], [], PORT_ATTRS), z[f"{tag}/snap_ports"])
                assert np.array_equal(env.query("vessels", [], [], VESSEL_ATTRS), z[f"{tag}/snap_vessels"])
                assert np.array_equal(env.query("matrices", [], [], MATRIX_ATTRS), z[f"{tag}/snap_matrices"])
    assert env.error == 0
    return z


def _make_oracle(topo, kwargs):
    return CimOracle(topo, **kwargs)


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_reference(name):
    replay_case(_make_oracle, name)


def test_reference_known_answers():
    """tests/cim/test_cim_scenario.py:156-204 (vessel periods), :239-261 (first arrivals),
    :297-324 (port table at first decision), :395-435 (early discharge), and the docs' full-length
    totals (docs/source/scenarios/container_inventory_management.rst:152-165, 293-303)."""
    z, meta = load_case("case_config_folder_kat")
    env = CimOracle(case_topology(meta), durations=200)
    hard_coded_period = [67, 75, 84, 67, 53, 58, 51, 58, 61, 49, 164, 182, 146, 164, 182, 146, 90, 98, 79, 95, 104,
                         84, 87, 97, 78, 154, 169, 136, 154, 169, 94, 105, 117, 94, 189, 210, 167, 189, 210, 167,
                         141, 158, 125, 141, 158, 125]
    assert env.vessel_period().tolist() == hard_coded_period
    met, dec, done = env.step(None)
    assert (dec[0], dec[2]) == (5, 35) and env.tick == 5
    assert dec[3] == 1240 and dec[4] == 0 and dec[5] == 0
    truth = [[223, 0, 14726], [16, 0, 916], [18, 0, 917], [89, 0, 5516], [84, 0, 4613], [72, 0, 4603], [26, 0, 1374],
             [24, 0, 1378], [48, 0, 2756], [54, 0, 2760], [26, 0, 1379], [99, 0, 5534], [137, 0, 7340], [19, 0, 912],
             [13, 0, 925], [107, 0, 6429], [136, 0, 9164], [64, 0, 3680], [24, 0, 1377], [31, 0, 1840],
             [109, 0, 6454], [131, 0, 7351]]
    got = env.query_live("ports", ["booking", "shortage", "empty"]).reshape(-1, 3)
    assert got.astype(int).tolist() == truth
    met, dec, done = env.step([(35, int(dec[1]), 1201, 0), (35, int(dec[1]), 1, 1)])
    assert (dec[0], dec[2]) == (6, 27)
    history = []
    while not done:
        if dec[2] == 35:
            v = env.query_live("vessels", ["full", "empty", "early_discharge"]).reshape(-1, 3)[35]
            history.append(tuple(int(x) for x in v))
        met, dec, done = env.step(None)
    assert history == [(465, 838, 362), (756, 547, 291), (1261, 42, 505), (1303, 0, 42), (1303, 0, 0), (1303, 0, 0),
                       (803, 0, 0)]
    for topo, expect in (("toy.4p_ssdd_l0.0", (2240000, 2190000, 0)), ("global_trade.22p_l0.0", (2240000, 1028481, 0))):
        env = CimOracle(topo, durations=1120)
        met, dec, done = env.step(None)
        while not done:
            met, dec, done = env.step(None)
        assert tuple(met.tolist()) == expect


class _OracleJoint:
    def __init__(self, topo, kwargs, mode):
        self.o, self.mode = CimOracle(topo, **kwargs), mode

    def step_joint(self, actions, n_answered):
        return self.o.step_joint(self.mode, actions, n_answered)

    def __getattr__(self, name):
        return getattr(self.o, name)


@pytest.mark.parametrize("name", joint_golden_cases())
def test_oracle_joint_decision_modes(name):
    """DecisionMode.Joint / JointWithSequentialAction (core.py:354-366) against the real reference."""
    replay_joint_case(_OracleJoint, name)


def test_real_data_known_answer():
    """tests/cim/test_cim_scenario.py:111-134 (test_load_from_real): port (booking, shortage, empty) at the first decision,
    identical for the csv and the binary form of the data set, and again after reset(keep_seed=True)."""
    truth = [[556, 0, 20751], [1042, 0, 17320], [0, 0, 25000], [0, 0, 25000]]
    for case in ("real_csv_rand0", "real_bin_none"):
        _, meta = load_case(case)
        env = CimOracle(case_topology(meta), durations=224 if case == "real_csv_rand0" else 100)
        for _ in range(2):
            env.step(None)
            assert env.query_live("ports", ["booking", "shortage", "empty"]).reshape(-1, 3).astype(int).tolist() == truth
            env.reset(keep_seed=True)
