"""A spread of the packaged topologies (every family, several noise levels) on the CPU wave emulator vs the oracle."""
import numpy as np
import pytest

from maro_amd.cim.topology import load_topology
from oracle.cim_oracle import CimOracle, hash_policy_action
from tests.backend_adapter import SingleEnvAdapter
from tests.emu.emu import EmuBackend
from tests.golden_util import MATRIX_ATTRS, PORT_ATTRS, VESSEL_ATTRS


@pytest.mark.parametrize("topology,seed", [("toy.4p_ssdd_l0.5", 3), ("toy.5p_ssddd_l0.2", 4096), ("toy.5p_ssddd_l0.8", 9),
                                           ("toy.6p_sssbdd_l0.4", 11), ("toy.6p_sssbdd_l0.7", 2), ("global_trade.22p_l0.6", 5)])
def test_topology_on_emulator(topology, seed):
    dur = 50
    topo = load_topology(topology)
    o = CimOracle(topo, durations=dur)
    o.set_seed(seed)
    o.reset(keep_seed=True)
    e = SingleEnvAdapter(EmuBackend(topo, 1, durations=dur, max_actions=1), seed=seed)
    om, od, odone = o.step(None)
    em, ed, edone = e.step(None)
    n = 0
    while not odone:
        assert not edone and np.array_equal(om, em) and np.array_equal(od, ed), (n, od, ed)
        a = hash_policy_action(seed, n, od)
        om, od, odone = o.step([a])
        em, ed, edone = e.step([a])
        n += 1
    assert edone and np.array_equal(om, em) and e.error == 0
    for node, attrs in (("ports", PORT_ATTRS), ("vessels", VESSEL_ATTRS), ("matrices", MATRIX_ATTRS)):
        assert np.array_equal(e.query(node, [], [], attrs), o.query(node, [], [], attrs)), node
