"""Fused observation (mrx_cim_set_observation): what step() writes equals snapshot_list slices of the decision frame."""
import numpy as np
import pytest

from maro_amd.cim.engine import PORT_ATTRS, VESSEL_ATTRS
from maro_amd.cim.topology import load_topology
from oracle.cim_oracle import hash_policy_action

P_ATTRS = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment", "transfer_cost"]
V_ATTRS = ["empty", "full", "remaining_space", "early_discharge"]


def check_fused_observation(backend, seeds, max_steps=400):
    """backend: EmuBackend-shaped (numpy)."""
    n = backend.n_envs
    topo = backend.topo
    pa, va = [PORT_ATTRS.index(a) for a in P_ATTRS], [VESSEL_ATTRS.index(a) for a in V_ATTRS]
    obs_p, obs_v = backend.set_observation(pa, va)
    backend.reset(np.asarray(seeds, np.int64))
    dec, met, done = backend.step()
    ports = np.arange(topo.n_ports, dtype=np.int32)
    step = fast = 0
    last_tick = np.full(n, -1)
    while not done.all() and step < max_steps:
        live = np.flatnonzero(dec[:, 7] == 1)
        qp = backend.query(0, dec[:, 6:7].copy(), ports, pa, len(pa))[:, 0]
        qv = backend.query(1, dec[:, 6:7].copy(), dec[:, 2:3].copy(), va, len(va))[:, 0, 0]
        op, ov = np.array(obs_p), np.array(obs_v)
        assert np.array_equal(op[live], qp[live]), step
        assert np.array_equal(ov[live], qv[live]), step
        fast += int((dec[live, 0] == last_tick[live]).sum())  # same tick as before: the HBM-direct fast path produced it
        last_tick = dec[:, 0].copy()
        acts = np.zeros((n, backend.max_actions, 4), np.int32)
        for e in live:
            acts[e, 0] = hash_policy_action(int(seeds[e]), step, dec[e])
        dec, met, done = backend.step(acts, (dec[:, 7] == 1).astype(np.int32), mask=(1 - done).astype(np.uint8))
        step += 1
    assert fast > 10 and step > 30
    return step


@pytest.mark.parametrize("topology", ["global_trade.22p_l0.8", "toy.5p_ssddd_l0.5"])
def test_fused_observation_matches_snapshot_slices(topology):
    from tests.emu.emu import EmuBackend
    b = EmuBackend(load_topology(topology), n_envs=3, durations=60, max_actions=1)
    check_fused_observation(b, seeds=[3, 4, 5])
