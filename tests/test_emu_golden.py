"""The engine's DEVICE SOURCE (maro_amd/csrc/cim_device.h), compiled for the host on a 64-fiber
wave emulator, replayed against the reference golden vectors.  Pure CPU: catches kernel-logic and
missing-sync bugs (forward vs reverse lane order) before any GPU time is spent.  The same replay
runs against the real HIP build in tests/test_gpu_golden.py."""
import pytest

from tests.backend_adapter import SingleEnvAdapter
from tests.emu.emu import EmuBackend
from tests.golden_util import golden_cases
from tests.test_oracle_golden import replay_case

FAST = [c for c in golden_cases() if "_full" not in c]


def _make(reverse, order_table=0):
    def make(topo, kwargs):
        kw = dict(kwargs)
        b = EmuBackend(topo, n_envs=1, start_tick=kw.get("start_tick", 0), durations=kw["durations"], snapshot_resolution=kw.get("snapshot_resolution", 1),
                       max_snapshots=kw.get("max_snapshots"), max_actions=2, reverse=reverse, order_table=order_table)
        assert b.layout.order_table_on == (1 if order_table >= 0 and topo.order_mode == 0 else 0)
        return SingleEnvAdapter(b)
    return make


@pytest.mark.parametrize("name", FAST)
def test_emulated_kernels_reproduce_reference(name):
    replay_case(_make(False), name)


@pytest.mark.parametrize("name", ["toy4p_l00_rand0", "gt22p_l08_res3", "toy6p_l08_rand0"])
def test_emulated_kernels_lane_order_independent(name):
    replay_case(_make(True), name)


@pytest.mark.parametrize("name", ["toy4p_l00_rand0", "gt22p_l08_rand0", "toy6p_l08_rand0", "gt22p_l08_reset_chain",
                                  "dump_case_config_kat"])
def test_online_order_generation_path(name):
    """order_table = -1: the step kernel draws each tick's orders itself (the only path for `unfixed` order mode)."""
    if name not in FAST:
        pytest.skip("golden not present")
    replay_case(_make(False, order_table=-1), name)


def test_real_data_needs_the_order_table():
    from tests.golden_util import case_topology, load_case
    _, meta = load_case("real_csv_rand0")
    with pytest.raises(RuntimeError, match="order table"):
        EmuBackend(case_topology(meta), n_envs=1, durations=50, order_table=-1)
