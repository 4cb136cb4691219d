"""mrx_cim_set_device_agent on the CPU wave emulator: the random legal agent answered INSIDE the step kernel (every launch form,
generic and plan-specialised build, with and without the fused observation) writes exactly the actions the separate policy
(oracle.cim_oracle.hash_policy_action = mrx_k_cim_random_policy's rule) would, so the trajectories are the oracle's."""
import numpy as np
import pytest

from maro_amd.cim.engine import PORT_ATTRS, VESSEL_ATTRS
from maro_amd.cim.topology import load_topology
from oracle.cim_oracle import CimOracle, hash_policy_action
from tests.emu.emu import EmuBackend

OBS = ([PORT_ATTRS.index(a) for a in ("empty", "full", "shortage")], [VESSEL_ATTRS.index(a) for a in ("empty", "remaining_space")])


@pytest.mark.parametrize("mode,obs", [(1, False), (2, True), (4, False), (5, True), ("S", True), ("S", False)])
@pytest.mark.parametrize("topology", ["toy.5p_ssddd_l0.5", "global_trade.22p_l0.8"])
def test_device_agent_equals_the_separate_policy(topology, mode, obs):
    topo = load_topology(topology)
    n, dur = 5, 40 if "22p" in topology else 60
    seeds = np.arange(n, dtype=np.int64) * 11 + 5
    kw = dict(n_envs=n, durations=dur, max_actions=1, max_snapshots=4)
    if mode == "S":
        kw.update(specialized=True, step_mode=2, spec_obs=OBS if obs else ((), ()))
    else:
        kw.update(step_mode=mode, pipe_waves=3)
    b = EmuBackend(topo, **kw)
    acts, nact, counts = np.zeros((n, 1, 4), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    b.set_device_agent(acts, nact, counts, next_key=1)
    if obs:
        b.set_observation(*OBS)          # (configured AFTER the agent: the agent's settings survive it)
    b.reset(seeds)
    oracles = []
    for e in range(n):
        o = CimOracle(topology, durations=dur)
        o.set_seed(int(seeds[e]))
        o.reset(keep_seed=True)
        oracles.append(o)
    ost = [o.step(None) for o in oracles]
    dec, met, done = b.step()
    k, answered = 0, np.zeros(n, np.int64)
    while not done.all():
        k += 1
        for e, (om, od, odone) in enumerate(ost):
            assert bool(done[e]) == bool(odone), (e, k)
            if odone:
                assert nact[e] == 0
                continue
            assert dec[e, :6].tolist() == [int(x) for x in od[:6]] and met[e].tolist() == [int(x) for x in om], (e, k)
            want = hash_policy_action(int(seeds[e]), k, od)
            assert nact[e] == 1 and acts[e, 0].tolist() == list(want), (e, k, acts[e, 0].tolist(), want)
            answered[e] += 1
            ost[e] = oracles[e].step([want])
        dec, met, done = b.step(acts, nact)   # the buffers the agent has just written
    assert all(s[2] for s in ost) and (counts == answered).all() and k > 10
    # switched off: the buffers are left alone
    b.set_device_agent()
    b.reset(seeds)
    acts[:] = -7
    b.step()
    assert (acts == -7).all()


def test_decision_keyed_draws_need_no_step_count():
    """next_key < 0: every draw is keyed on its decision's (tick, vessel) — the form a captured launch sequence uses."""
    topo = load_topology("toy.5p_ssddd_l0.5")
    n, dur = 3, 50
    seeds = np.arange(n, dtype=np.int64) + 77
    a, b = EmuBackend(topo, n_envs=n, durations=dur, max_actions=1, step_mode=1), EmuBackend(topo, n_envs=n, durations=dur, max_actions=1, step_mode=2)
    bufs = [(np.zeros((n, 1, 4), np.int32), np.zeros(n, np.int32)) for _ in range(2)]
    a.set_device_agent(*bufs[0], None, next_key=-1)
    b.set_device_agent(*bufs[1], None, next_key=-1)
    a.reset(seeds), b.reset(seeds)
    ra, rb = a.step(), b.step()
    steps = 0
    while not ra[2].all():
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb)) and np.array_equal(bufs[0][0], bufs[1][0]) and np.array_equal(bufs[0][1], bufs[1][1])
        ra, rb = a.step(*bufs[0]), b.step(*bufs[1])
        steps += 1
    assert steps > 20
