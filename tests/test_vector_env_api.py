"""Host-side object API (GpuVectorEnv / GpuEnvView / payloads / snapshot slicing) against the oracle.
Runs on the CPU wave emulator here; tests/test_gpu_vector_env.py runs the same checks on the HIP engine."""
import pickle

import numpy as np
import pytest

from maro_amd.cim.payloads import Action, ActionType, DecisionEvent
from maro_amd.cim.vector_env import BackendsInvalidAttributeException, GpuVectorEnv, InvalidActionError
from oracle.cim_oracle import CimOracle

TOPO = "toy.5p_ssddd_l0.6"


def make_env(batch, engine_factory, **kw):
    eng = engine_factory(TOPO, batch, max_actions=2, **kw)
    return GpuVectorEnv(batch, "cim", TOPO, durations=kw.get("durations", 100), _engine=eng)


def emu_factory(topology, n, **kw):
    from tests.emu.emu_engine import EmuEngine
    return EmuEngine(topology, n, **kw)


def check_vector_env(engine_factory):
    env = make_env(3, engine_factory, durations=60)
    oracles = [CimOracle(TOPO, durations=60) for _ in range(3)]
    ost = [o.step(None) for o in oracles]
    metrics, events, all_done = env.step(None)
    assert not all_done and len(events) == 3
    step = 0
    while not all_done:
        actions = []
        for e, (ev, (om, od, odone)) in enumerate(zip(events, ost)):
            if ev is None:
                assert odone
                actions.append(None)
                continue
            assert isinstance(ev, DecisionEvent)
            assert (ev.tick, ev.port_idx, ev.vessel_idx, ev.action_scope.load, ev.action_scope.discharge,
                    ev.early_discharge) == tuple(int(x) for x in od[:6])
            assert metrics[e]["order_requirements"] == om[0] and metrics[e]["container_shortage"] == om[1]
            # env 0: no action, env 1: single Action, env 2: list of two actions
            if e == 0:
                actions.append(None)
                ost[e] = oracles[e].step(None)
            elif e == 1:
                q = ev.action_scope.discharge // 2
                actions.append(Action(ev.vessel_idx, ev.port_idx, q, ActionType.DISCHARGE))
                ost[e] = oracles[e].step([(ev.vessel_idx, ev.port_idx, q, 1)])
            else:
                q = ev.action_scope.load // 3
                actions.append([Action(ev.vessel_idx, ev.port_idx, q, ActionType.LOAD),
                                Action(ev.vessel_idx, ev.port_idx, 0, ActionType.DISCHARGE)])
                ost[e] = oracles[e].step([(ev.vessel_idx, ev.port_idx, q, 0), (ev.vessel_idx, ev.port_idx, 0, 1)])
            if step == 3 and e == 2:
                # snapshot slicing while paused: current frame == live state; past frames; missing frame -> zeros
                ev2 = pickle.loads(pickle.dumps(ev))
                assert ev2.action_scope.load == ev.action_scope.load
        metrics, events, all_done = env.step(actions)
        step += 1
    for e, (om, od, odone) in enumerate(ost):
        assert odone and metrics[e]["operation_number"] == om[2]
    # after the end: (None, None, True), vector_env/env_process.py:34-37
    m, ev, d = env.step(None)
    assert d and all(x is None for x in m) and all(x is None for x in ev)
    # full-history queries, list of flat float64 arrays (np_backend.pyx:520-549)
    got = env.snapshot_list["ports"][::["empty", "full", "shortage", "transfer_cost"]]
    for e in range(3):
        exp = oracles[e].query("ports", [], [], ["empty", "full", "shortage", "transfer_cost"])
        assert got[e].dtype == np.float64 and np.array_equal(got[e], exp)
    got = env.snapshot_list["vessels"][[3, 59, 1000]:[0, 2]:("remaining_space", "future_stop_tick_list")]
    for e in range(3):
        assert np.array_equal(got[e], oracles[e].query("vessels", [3, 59, 1000], [0, 2], ["remaining_space", "future_stop_tick_list"]))
    assert env.snapshot_list["ports"][0:0:None] is None
    with pytest.raises(BackendsInvalidAttributeException):
        env.snapshot_list["ports"][0:0:"nope"]
    assert env.snapshot_list.get_frame_index_list()[0] == oracles[0].frame_indices()
    assert env.tick == [o.tick for o in oracles]
    assert len(env.snapshot_list["vessels"]) == oracles[0].topo.n_vessels


def check_env_view(engine_factory):
    env = make_env(2, engine_factory, durations=50)
    view = env.env_view(1)
    o = CimOracle(TOPO, durations=50)
    # dict-addressed stepping leaves env 0 untouched (vector_env.py:199-209)
    m, ev, d = view.step(None)
    om, od, odone = o.step(None)
    assert env.tick[0] == 0 and view.tick == o.tick == ev.tick
    assert view.frame_index == od[6] and view.agent_idx_list == list(range(o.topo.n_ports))
    # paused: the current frame is visible through snapshot_list (pre-decision snapshot, core.py:345)
    cur = view.snapshot_list["ports"][view.frame_index::["empty", "booking"]]
    assert np.array_equal(cur, o.query("ports", [int(od[6])], [], ["empty", "booking"]))
    assert view.snapshot_list.get_frame_index_list() == o.frame_indices()
    # env.business_engine (abs_core.py:71): the read-only view callers use for frame / snapshots / metrics / agent list
    be = view.business_engine
    assert be.snapshots is view.snapshot_list and be.frame.snapshots is view.snapshot_list
    assert be.get_metrics() == view.metrics and be.get_agent_idx_list() == view.agent_idx_list
    assert be.frame_index(view.tick) == view.frame_index and be.scenario_name == "cim" and be.configs is view.configs
    assert set(be.get_node_mapping()) == {"ports", "vessels"}
    mapping = view.get_ticks_frame_index_mapping()
    assert mapping[view.tick] == view.frame_index and sorted(set(mapping.values())) == view.snapshot_list.get_frame_index_list()
    with pytest.raises(InvalidActionError):
        view.step(Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge + 1, ActionType.DISCHARGE))
    # seed protocol: set_seed + reset(keep_seed=True) -> explicit seed; reset() -> redraw from the route stream
    for keep, seed in ((True, 7), (False, None), (True, None)):
        if seed is not None:
            view.set_seed(seed)
            o.set_seed(seed)
        view.reset(keep_seed=keep)
        o.reset(keep_seed=keep)
        m, ev, d = view.step(None)
        om, od, odone = o.step(None)
        n = 0
        while not d:
            assert (ev.tick, ev.port_idx, ev.vessel_idx) == tuple(int(x) for x in od[:3])
            m, ev, d = view.step(None)
            om, od, odone = o.step(None)
            n += 1
        assert odone and m["container_shortage"] == om[1] and view.metrics["order_requirements"] == om[0] and n > 3
    assert view.step(None) == (None, None, True)


def test_vector_env_on_emulator():
    check_vector_env(emu_factory)


def test_env_view_on_emulator():
    check_env_view(emu_factory)


def check_joint_object_api(engine_factory, case):
    """DecisionMode.Joint / JointWithSequentialAction through GpuVectorEnv: lists of DecisionEvents in, lists of Actions
    out, replaying what the real reference did (tests/golden/cimjoint_*.npz)."""
    from tests.golden_util import case_topology, load_joint_case
    z, meta = load_joint_case(case)
    topo = case_topology(meta)
    kw = meta["kwargs"]
    mode = meta["decision_mode"]
    eng = engine_factory(topo, 2, max_actions=topo.n_vessels, decision_mode=mode, **kw)
    env = GpuVectorEnv(2, "cim", topo, decision_mode=mode, _engine=eng, **kw)
    gd, ga, gn, gm = z["decisions"], z["actions"], z["n_answered"], z["metrics"]
    metrics, events, all_done = env.step(None)
    i = 0
    while not all_done:
        for e in range(2):
            evs = events[e]
            want = gd[i][gd[i][:, 7] == 1]
            assert isinstance(evs, list) and len(evs) == len(want)
            got = [[ev.tick, ev.port_idx, ev.vessel_idx, ev.action_scope.load, ev.action_scope.discharge, ev.early_discharge] for ev in evs]
            assert got == want[:, :6].tolist(), (case, i, e)
            assert [metrics[e][k] for k in ("order_requirements", "container_shortage", "operation_number")] == gm[i].tolist()
        k = int(gn[i])
        acts = [Action(int(a[0]), int(a[1]), int(a[2]), ActionType.LOAD if a[3] == 0 else ActionType.DISCHARGE) for a in ga[i][:k]]
        # env 0: one Action per event; env 1: the same wrapped in single-element lists (both forms are accepted, core.py:301-313)
        metrics, events, all_done = env.step([acts, [[a] for a in acts]])
        i += 1
    assert i == len(gd)
    assert [metrics[0][k] for k in ("order_requirements", "container_shortage", "operation_number")] == z["final_metrics"].tolist()
    snap = env.snapshot_list["ports"][::["empty", "full", "shortage", "booking"]]
    assert np.array_equal(snap[0], snap[1])


@pytest.mark.parametrize("case", ["jointseq_toy5p_l05_some", "joint_toy6p_l08_all"])
def test_joint_decision_modes_object_api(case):
    check_joint_object_api(emu_factory, case)


def check_invalid_action_in_a_batch(engine_factory):
    """An invalid action (the reference asserts, cim/business_engine.py:731,736) in ONE env of a batch: the error names it,
    the other envs' results of that step are kept, only the INVALID_ACTION status bit is cleared, and stepping goes on."""
    env = make_env(3, engine_factory, durations=60)
    oracles = [CimOracle(TOPO, durations=60) for _ in range(3)]
    ost = [o.step(None) for o in oracles]
    metrics, events, all_done = env.step(None)
    bad = Action(events[1].vessel_idx, events[1].port_idx, events[1].action_scope.discharge + 5, ActionType.DISCHARGE)
    with pytest.raises(InvalidActionError, match=r"\[1\]"):
        env.step([None, bad, None])
    ost = [o.step(None) for o in oracles]   # the engine skipped the bad action: same as no action
    assert int(env.engine.status[1]) & 1 == 0
    metrics, events, all_done = env.step(None)
    ost = [o.step(None) for o in oracles]
    for e, (om, od, odone) in enumerate(ost):
        assert (events[e].tick, events[e].vessel_idx) == (int(od[0]), int(od[2])) and metrics[e]["order_requirements"] == om[0]
    # reset: fresh metrics (Env.reset), not the previous episode's
    view = env.env_view(2)
    assert view.metrics["order_requirements"] > 0
    view.reset(keep_seed=True)
    assert view.metrics == {"order_requirements": 0, "container_shortage": 0, "operation_number": 0}


def test_invalid_action_in_a_batch_on_emulator():
    check_invalid_action_in_a_batch(emu_factory)


def test_unknown_scenario_dispatches_to_the_reference_vector_env(monkeypatch):
    """a scenario / business engine the GPU engines do not implement runs on maro.vector_env.VectorEnv when
    MARO is importable; otherwise the constructor says so."""
    import sys
    import types
    monkeypatch.setitem(sys.modules, "maro", None)   # not importable
    with pytest.raises(NotImplementedError, match="not importable"):
        GpuVectorEnv(2, "vm_scheduling", topology="azure.2019.10k", durations=10)
    seen = {}

    class FakeVectorEnv:
        def __init__(self, batch_num, **kw):
            seen.update(batch_num=batch_num, **kw)
    pkg, mod = types.ModuleType("maro"), types.ModuleType("maro.vector_env")
    mod.VectorEnv = FakeVectorEnv
    pkg.vector_env = mod
    monkeypatch.setitem(sys.modules, "maro", pkg)
    monkeypatch.setitem(sys.modules, "maro.vector_env", mod)
    env = GpuVectorEnv(2, "vm_scheduling", topology="azure.2019.10k", durations=10, device="cuda:0")
    assert isinstance(env, FakeVectorEnv) and seen == dict(batch_num=2, scenario="vm_scheduling", topology="azure.2019.10k", durations=10)
    env = GpuVectorEnv(1, "cim", topology="toy.4p_ssdd_l0.0", durations=10, business_engine_cls=object)
    assert isinstance(env, FakeVectorEnv) and seen["business_engine_cls"] is object
    # positional arguments after `scenario` are bound by name (a positional topology must not land in the scenario slot),
    # and an int decision_mode becomes the reference's enum
    dm = types.ModuleType("maro.simulator.utils.common")

    class DecisionMode:
        Sequential, Joint, JointWithSequentialAction = "seq", "joint", "jwsa"
    dm.DecisionMode = DecisionMode
    for name in ("maro.simulator", "maro.simulator.utils"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "maro.simulator.utils.common", dm)
    seen.clear()
    env = GpuVectorEnv(2, "vm_scheduling", "azure.2019.10k", 0, 25, decision_mode=1, seeds=[1, 2])
    assert isinstance(env, FakeVectorEnv)
    assert seen == dict(batch_num=2, scenario="vm_scheduling", topology="azure.2019.10k", start_tick=0, durations=25, decision_mode="joint")
    with pytest.raises(TypeError, match="multiple values"):
        GpuVectorEnv(2, "vm_scheduling", "azure.2019.10k", topology="x")


# ---- the object API over a stream-pipelined batch (PipelinedCimBatch as GpuVectorEnv's engine: groups of envs, here 3 + 2 + 2
# emulator engines; tests/test_gpu_vector_env.py runs the same on HIP engines with their streams)
def pipelined_emu_factory(groups):
    def factory(topology, n, **kw):
        from maro_amd.cim.rollout import PipelinedCimBatch
        return PipelinedCimBatch(topology, n, groups=groups, device="cpu", engine_factory=emu_factory, seeds="topology", **kw)
    return factory


def check_pipelined_object_api(single_factory, grouped_factory, n=7, steps=60):
    """A grouped batch behaves exactly like one engine: dict / list actions, an invalid action, seeds, reset, snapshot slices."""
    import random
    envs = [GpuVectorEnv(n, "cim", TOPO, durations=80, _engine=f(TOPO, n, max_actions=2, durations=80)) for f in (single_factory, grouped_factory)]
    rng = random.Random(5)

    def key(ev):
        return None if ev is None else (ev.tick, ev.port_idx, ev.vessel_idx, ev.action_scope.load, ev.action_scope.discharge)

    def same(ra, rb):
        assert ra[0] == rb[0] and ra[2] == rb[2] and [key(e) for e in ra[1]] == [key(e) for e in rb[1]]

    last = [env.step(None) for env in envs]
    same(*last)
    assert envs[0].env_view(0).business_engine.calc_max_snapshots() == envs[1].env_view(n - 1).business_engine.calc_max_snapshots() > 0   # (grouped: PipelinedCimBatch.layout)
    cur = list(last[0][1])                  # the latest decision event of every env
    for k in range(steps):
        if last[0][2]:
            break
        acts = [None if ev is None else Action(ev.vessel_idx, ev.port_idx, rng.randint(0, ev.action_scope.load), ActionType.LOAD) for ev in cur]
        if k % 5 == 4:      # dict form: a subset of the envs steps (one or two per group)
            sub = sorted(e for e in range(n) if e % 2 == 0)
            last = [env.step({e: acts[e] for e in sub}) for env in envs]
            for e, ev in zip(sub, last[0][1]):
                cur[e] = ev
        else:
            last = [env.step(list(acts)) for env in envs]
            cur = list(last[0][1])
        same(*last)
        q = [env.snapshot_list["ports"][::["empty", "full", "shortage"]] for env in envs]
        for x, y in zip(*q):
            assert np.array_equal(x, y)
        assert envs[0].tick == envs[1].tick and envs[0].frame_index == envs[1].frame_index
    for env in envs:
        for e in range(n):
            env.set_seed(11 + e, [e])
        env.reset(keep_seed=True)
    a, b = (env.step(None) for env in envs)
    same(a, b)
    assert len({key(e) for e in a[1]}) > 1      # different seeds: the envs really diverge
    # an invalid action in the LAST group is reported for that env only and acknowledged in that group's status word
    ev = a[1][n - 1]
    bad = {n - 1: Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge + 5, ActionType.DISCHARGE)}
    for env in envs:
        with pytest.raises(InvalidActionError, match=str(n - 1)):
            env.step(dict(bad))
        assert int(env.engine.status[n - 1]) & 1 == 0


def test_object_api_on_a_pipelined_batch_equals_a_single_engine():
    check_pipelined_object_api(emu_factory, pipelined_emu_factory(3))


def check_whole_batch_step_equals_the_per_env_path(engine_factory):
    """step(list) of the whole batch takes the array-form path (_step_all), step(dict) the per-env path (_step_envs): same
    metrics, events, done flags, ticks and frame indices step by step; `tick` / `frame_index` come from the step's own read-back."""
    n = 5
    a, b = make_env(n, engine_factory, durations=50), make_env(n, engine_factory, durations=50)
    ra, rb = a.step(None), b.step({e: None for e in range(n)})
    k = 0
    while True:
        (ma, ea, da), (mb, eb, db) = ra, rb
        assert da == db and ma == mb and a.tick == b.tick and a.frame_index == b.frame_index
        assert a.tick == a.engine.ticks.cpu().tolist()
        for x, y in zip(ea, eb):
            assert (x is None) == (y is None)
            if x is not None:
                assert (x.tick, x.port_idx, x.vessel_idx, x.action_scope.load, x.action_scope.discharge, x.early_discharge) == \
                       (y.tick, y.port_idx, y.vessel_idx, y.action_scope.load, y.action_scope.discharge, y.early_discharge)
                assert x.snapshot_list["ports"][x.tick:x.port_idx:"empty"].tolist() == y.snapshot_list["ports"][y.tick:y.port_idx:"empty"].tolist()
        if da:
            break
        acts = [None if ev is None or (e + k) % 3 == 0 else Action(ev.vessel_idx, ev.port_idx, ev.action_scope.load // 2, ActionType.LOAD) if (e + k) % 3 == 1
                else [Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge // 2, ActionType.DISCHARGE)] for e, ev in enumerate(ea)]
        ra, rb = a.step(acts), b.step(dict(enumerate(acts)))
        k += 1
    assert k > 10
    # a finished batch: (None, None, True) for every env, in both forms; a single broadcast Action
    assert a.step(None) == ([None] * n, [None] * n, True) and b.step({0: None})[0] == [None]
    a.reset(), b.reset()
    ra, rb = a.step(None), b.step({e: None for e in range(n)})
    one = Action(ra[1][0].vessel_idx, ra[1][0].port_idx, 0, ActionType.DISCHARGE)   # (every env is at the same first decision)
    ra, rb = a.step(one), b.step({e: one for e in range(n)})
    assert ra[0] == rb[0] and a.tick == b.tick


def test_whole_batch_step_equals_the_per_env_path_on_emulator():
    check_whole_batch_step_equals_the_per_env_path(emu_factory)


def test_c_object_builders_equal_the_python_comprehensions(monkeypatch):
    """maro_amd/_fastobj.so (csrc_host/fastobj.c: actions encoded, DecisionEvents and metrics dicts built in C) and the pure-Python
    form of the whole-batch step give the same lists; events pickle and print like the reference's; foreign action enums and
    numpy integers are accepted by the C encoder as by the Python one."""
    import enum

    from maro_amd.cim import vector_env as ve
    if ve._FO is None:
        pytest.skip("maro_amd/_fastobj.so not built")
    n = 4
    a, b = make_env(n, emu_factory, durations=40), make_env(n, emu_factory, durations=40)
    ra = a.step(None)
    monkeypatch.setattr(ve, "_FO", None)
    rb = b.step(None)
    monkeypatch.undo()

    class Foreign(enum.Enum):
        LOAD = "l"
        DISCHARGE = "d"

    k = 0
    while not ra[2]:
        assert ra[0] == rb[0] and ra[2] == rb[2]
        for x, y in zip(ra[1], rb[1]):
            assert (x is None) == (y is None)
            if x is not None:
                assert repr(x) == repr(y) and repr(pickle.loads(pickle.dumps(x))) == repr(pickle.loads(pickle.dumps(y))) == repr(x)
                assert (x.tick, x.port_idx, x.vessel_idx, x.early_discharge) == (y.tick, y.port_idx, y.vessel_idx, y.early_discharge)
        acts = []
        for e, ev in enumerate(ra[1]):
            if ev is None or (e + k) % 4 == 0:
                acts.append(None)
            elif (e + k) % 4 == 1:
                acts.append(Action(np.int64(ev.vessel_idx), np.int32(ev.port_idx), np.int64(ev.action_scope.load // 2), Foreign.LOAD))
            elif (e + k) % 4 == 2:
                acts.append([Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge // 3, ActionType.DISCHARGE)])
            else:
                acts.append(Action(ev.vessel_idx, ev.port_idx, 0, ActionType.LOAD))
        ra = a.step(acts)
        monkeypatch.setattr(ve, "_FO", None)
        rb = b.step(list(acts))
        monkeypatch.undo()
        k += 1
    assert k > 8 and ra[0] == rb[0]
